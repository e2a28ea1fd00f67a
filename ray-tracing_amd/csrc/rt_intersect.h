// rt_intersect.h -- event counters, the torus solve, the candidate tests and the forms of trace:
// the per-object loop (trace_closest), the sphere hierarchy with every lane walking its own lists
// (trace_hierarchy), the same with the wave's work pooled in LDS (trace_hierarchy_flat, what the
// large-scene kernels run) and one ray at a time on all lanes (trace_cooperative, the last rays of
// a wave in deep frames).  All return the reference loop's (t, id) bit for bit.  Included by
// rt_device.hip after rt_numerics.h; not a stand-alone header.
#pragma once

namespace RT_NS {

// ------------------------------------------------------------------ event counters
// Two uses of the same LDS counter array in the instrumented flavour: event counts (one per
// lane and event, the FLOP accounting), or -- wave_mode -- a profile of the code blocks a wave
// executes (WCNT: once per wave that enters the block with any lane; LCNT: lanes in it).
enum { HS_TOP, HS_LEAF, HS_CAND, HS_SCAN, HS_COOP_RAY, HS_COOP_ROUND,
       // development statistics of the pooled trace (lane sums only)
       HS_RAYS, HS_RAYS_SHADOW, HS_RAYS_HIT, HS_KEPT, HS_KEPT_NEEDED, HS_KEPT_SHADOW_NEEDED, HS_CANDS, HS_CANDS_NEEDED,
       HS_TORUS_ROUND, HS_RAYS_NEAR_HIT,
       // development timers (core clock cycles per wave, as wave-level sums): loop head + refill, cooperative trace,
       // node tests + item lists, leaf rounds, exact rounds, owner + replay, shading
       HS_T_HEAD, HS_T_COOP, HS_T_SCAN, HS_T_LEAF, HS_T_EXACT, HS_T_OWNER, HS_T_SHADE,
       HS_COUNT };
template <bool DEBUG> struct Counters {
    unsigned int *lds_ev;
    DEV void add(int) const {}
    DEV void wave(int) const {}
    DEV void lanes(int) const {}
    DEV bool counting_events() const { return false; }
#ifdef RT_DEV_SWITCHES
#ifndef RT_DEV_TIMERS_ONLY
    // development builds: loop iterations of the hierarchy walk, per wave (the lowest active lane counts) and per lane
    mutable unsigned int hs_wave[HS_COUNT] = {}, hs_lane[HS_COUNT] = {};
    DEV void hstat(int k) const
    {
        hs_lane[k]++;
        if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) hs_wave[k]++;
    }
    DEV void hadd(int k, unsigned int n) const { hs_lane[k] += n; }
#else
    // (-DRT_DEV_TIMERS_ONLY: the phase timers alone; the per-lane counters cost the kernels so many registers that
    // the shares of the phases no longer are those of the product build)
    DEV void hstat(int) const {}
    DEV void hadd(int, unsigned int) const {}
#endif
    // Phase timers: tick(k) charges the cycles since the last tick to phase k (wave-uniform values).  Coarse
    // phases only: every tick is a scalar memory instruction the wave waits for, and the compiler may move
    // arithmetic across it, so shares of short stretches of code are not to be trusted.
    mutable unsigned long long t_last = 0, hs_time[HS_COUNT - HS_T_HEAD] = {};
    DEV void tick0() const { t_last = __builtin_amdgcn_s_memtime(); }
    DEV void tick(int k) const { const unsigned long long t = __builtin_amdgcn_s_memtime(); hs_time[k - HS_T_HEAD] += t - t_last; t_last = t; }
#else
    DEV void tick0() const {}
    DEV void tick(int) const {}
    DEV void hstat(int) const {}
#endif
};
template <> struct Counters<true> {
    unsigned int *lds_ev;
    int wave_mode;
    DEV void add(int ev) const { if (!wave_mode) atomicAdd(&lds_ev[ev], 1u); }
    DEV void wave(int slot) const
    {
        if (wave_mode) {
            unsigned long long m = __ballot(1);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&lds_ev[slot], 1u);
        }
    }
    DEV void lanes(int slot) const { if (wave_mode) atomicAdd(&lds_ev[slot], 1u); }
    // the event counts are the reference's work, object by object: no shortcuts while counting
    DEV bool counting_events() const { return !wave_mode; }
    DEV void hstat(int) const {}
    DEV void tick0() const {}
    DEV void tick(int) const {}
};
#define CNT(ev) cnt.add(ev)
#define WCNT(slot) cnt.wave(slot)
#define LCNT(slot) cnt.lanes(slot)
#define HSTAT(k) cnt.hstat(k)

// ------------------------------------------------------------------ torus (fragment_shader.glsl:250-375)
#define Q_EPS 1e-7f

// [census: root_offer @ QUAD_ROOTS*0.2]   (ten inlined copies; a quadratic with real roots runs two of them)
struct RootMin {
    float t; bool found;
    DEV void offer(float r) { if (r > 0.001f && r < t) { t = r; found = true; } }
};

// solve_quadratic (:159-180) for the monic calls of solve_quartic: a = 1.
// bb is b*b as the compiled reference has it.  Offers (root - offset) to `rm`.
template <bool DEBUG>
DEV void quad_monic_offer(float bb, float b, float c, float offset, RootMin &rm, const Counters<DEBUG> &cnt)
{
    // [census: quad_monic @ QUAD_OFFER]
    WCNT(WP_QUAD_OFFER);
    float disc = bb - 4.0f * c;
    if (disc < -Q_EPS) { CNT(EV_QUAD_EARLY); return; }
    // [census: quad_monic_roots @ QUAD_ROOTS]
    CNT(EV_QUAD_FULL);
    WCNT(WP_QUAD_ROOTS);
    disc = max0(disc);
    float sq = rt_sqrt(disc);
    CNT(EV_ROOT_OUT);
    rm.offer((-b + sq) * 0.5f - offset);
    if (!(disc < Q_EPS)) { CNT(EV_ROOT_OUT); rm.offer((-b - sq) * 0.5f - offset); }
}

// Smallest root > 0.001 of t^4 + A t^3 + B t^2 + C t + D (Ferrari, :250-305 + :345-352).
template <bool DEBUG>
DEV bool quartic_nearest(float a, float b, float c, float d, float &t_out, const Counters<DEBUG> &cnt)
{
    // [census: quartic_depress @ TORUS]
    float a2 = a * a;
    float p = b - 0.375f * a2;
    float q = (c - (0.5f * a) * b) + (0.125f * a2) * a;
    float r = ((d - (0.25f * a) * c) + (0.0625f * a2) * b) - (0.01171875f * a2) * a2;
    float offset = 0.25f * a;
    RootMin rm; rm.t = 1e20f; rm.found = false;
    if (__builtin_fabsf(q) < Q_EPS) {
        // biquadratic (:262-276): quadratic in y^2
        // [census: biquadratic @ BIQUADRATIC]
        WCNT(WP_BIQUADRATIC);
        float disc = p * p - 4.0f * r;
        if (disc < -Q_EPS) { CNT(EV_QUAD_EARLY); }
        else {
            CNT(EV_QUAD_FULL);
            disc = max0(disc);
            float sq = rt_sqrt(disc);
            int ny = disc < Q_EPS ? 1 : 2;
            float z0 = (-p + sq) * 0.5f, z1 = (-p - sq) * 0.5f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float z = i == 0 ? z0 : z1;
                if (i < ny && z >= -Q_EPS) {
                    CNT(EV_BIQ_Z);
                    float y = rt_sqrt(max0(z));
                    CNT(EV_ROOT_OUT);
                    rm.offer(y - offset);
                    if (y > Q_EPS) { CNT(EV_ROOT_OUT); rm.offer(-y - offset); }
                }
            }
        }
    } else {
        // resolvent cubic u^3 + 2p u^2 + (p^2 - 4r) u - q^2 (:281), first root only (:285)
        // [census: cubic_setup @ CUBIC]
        CNT(EV_CUBIC);
        WCNT(WP_CUBIC);
        float an = 2.0f * p;
        float bn = p * p - 4.0f * r;
        float cn = -(q * q);
        float an2 = an * an;
        float pc = bn - rt_div_const<3>(an2);
        float off_c = rt_div_const<3>(an);
        float qc = (cn - rt_div_const<3>(an * bn)) + rt_div_const<27>((2.0f * an2) * an);
        float half_q = 0.5f * qc;
        float p3 = (pc * pc) * pc;
        float disc = half_q * half_q + rt_div_const<27>(p3);
        float u;
        if (disc >= -Q_EPS) {
            // [census: cardano @ CARDANO]
            CNT(EV_CARDANO);
            WCNT(WP_CARDANO); LCNT(WP_CARDANO_LANES);
            disc = max0(disc);
            float sq = rt_sqrt(disc);
            float A = -half_q + sq;
            float B = -half_q - sq;
            float rootA = sign_f(A) * rt_pow(__builtin_fabsf(A), 0x1.555556p-2f);
            float rootB = sign_f(B) * rt_pow(__builtin_fabsf(B), 0x1.555556p-2f);
            u = (rootA - off_c) + rootB;
        } else {
            // [census: trig @ TRIG]
            CNT(EV_TRIG);
            WCNT(WP_TRIG); LCNT(WP_TRIG_LANES);
            float rho = rt_sqrt(rt_div_const<27>(-p3));
            float x = rt_div(-half_q, rho);
            x = (x > -1.0f) ? x : -1.0f;       // NaN -> -1
            x = (x < 1.0f) ? x : 1.0f;
            float theta = rt_div_const<3>(rt_acos(x));
            float m = 2.0f * rt_sqrt(rt_div_const<3>(-pc));
            u = m * rt_cos(theta) - off_c;
        }
        // [census: ferrari_tail @ FERRARI_TAIL]
        if (!(u < 0.0f)) {
            CNT(EV_FERRARI_TAIL);
            WCNT(WP_FERRARI_TAIL);
            float w = rt_sqrt(u);
            float term_A = 0.5f * (p + u);
            float term_B = rt_div(0.5f * q, w + Q_EPS);
            float bb = __builtin_fabsf(u);
            quad_monic_offer<DEBUG>(bb, w, term_A - term_B, offset, rm, cnt);
            quad_monic_offer<DEBUG>(bb, -w, term_A + term_B, offset, rm, cnt);
        }
    }
    // [census: quartic_end @ TORUS]
    t_out = rm.t;
    return rm.found;
}

struct TorusFrame { float ux, uy, uz, wx, wy, wz, vx, vy, vz, R, R2, r2, fourR2, eightR2, r; };

DEV TorusFrame load_torus(const float *lds, int off_torus, int slot)
{
    const float4 *q = reinterpret_cast<const float4 *>(lds + off_torus + 16 * slot);
    float4 a = q[0], b = q[1], c = q[2], d = q[3];
    TorusFrame f;
    f.ux = a.x; f.uy = a.y; f.uz = a.z; f.wx = a.w;
    f.wy = b.x; f.wz = b.y; f.vx = b.z; f.vy = b.w;
    f.vz = c.x; f.R = c.y; f.R2 = c.z; f.r2 = c.w;
    f.fourR2 = d.x; f.eightR2 = d.y; f.r = d.z;
    return f;
}

// Newton polish of a torus root (not part of the reference, whose solve ends with Ferrari's roots,
// :345-352; RT_FLAG_NEWTON_POLISH).  The quartic's coefficients are differences of terms of size
// |ro|^4, so in binary32 its roots are off by 1e-4..1e-2 of t and some are no intersections at
// all.  The polish leaves the polynomial and works on the torus itself, in its own frame:
// g(p) = (rho - R)^2 + y^2 - r^2, rho = sqrt(x^2 + z^2), is zero on the surface and well
// conditioned near it.  Five Newton steps on g(ro + t rd), each at most r long; the best of the
// six points visited replaces the root if it is past 0.001 and within 4r of the start (it cannot
// wander to another crossing of the tube), otherwise the root stays.  A root whose best |g| is
// still above 0.01 r^2 does not lie on the torus: the hit is dropped.  Plain binary32 operations
// in the order of the CPU oracle's torus_polish, so the strict flavour equals it bit for bit.
DEV float torus_g(float rox, float roy, float roz, float rdx, float rdy, float rdz, float R, float r2, float t, float &dg)
{
    float px = rox + t * rdx, py = roy + t * rdy, pz = roz + t * rdz;
    float rho = rt_sqrt(px * px + pz * pz);
    float a = rho - R;
    float g = (a * a + py * py) - r2;
    dg = 2.0f * (a * rt_div(px * rdx + pz * rdz, rho) + py * rdy);
    return g;
}
DEV bool torus_polish(float rox, float roy, float roz, float rdx, float rdy, float rdz, float R, float r, float &t_io)
{
    const float r2 = r * r, t0 = t_io;
    float t = t0, dg;
    float g = torus_g(rox, roy, roz, rdx, rdy, rdz, R, r2, t, dg);
    float best_t = t0, best_g = __builtin_fabsf(g);
    for (int it = 0; it < 5; ++it) {
        if (!(__builtin_fabsf(dg) > 0.0f)) break;                 // zero or NaN slope
        float step = rt_div(g, dg);
        step = step > r ? r : step;
        step = step < -r ? -r : step;
        t = t - step;
        g = torus_g(rox, roy, roz, rdx, rdy, rdz, R, r2, t, dg);
        if (__builtin_fabsf(g) < best_g) { best_g = __builtin_fabsf(g); best_t = t; }
    }
    if (best_t > 0.001f && __builtin_fabsf(best_t - t0) <= 4.0f * r) t_io = best_t;
    else best_g = __builtin_fabsf(torus_g(rox, roy, roz, rdx, rdy, rdz, R, r2, t0, dg));
    return best_g <= 0.01f * r2;                                  // false for NaN
}

// intersect_torus (:309-352) up to the root; ocx.. = ray origin - centre.
template <bool DEBUG>
DEV bool torus_nearest(const TorusFrame &f, float ocx, float ocy, float ocz, float dx, float dy, float dz,
                       int polish, float &t_out, const Counters<DEBUG> &cnt)
{
    // [census: torus_setup @ TORUS]
    CNT(EV_TORUS_CALL);
    float rox = dot3(ocx, ocy, ocz, f.ux, f.uy, f.uz);
    float roy = dot3(ocx, ocy, ocz, f.wx, f.wy, f.wz);
    float roz = dot3(ocx, ocy, ocz, f.vx, f.vy, f.vz);
    float rdx = dot3(dx, dy, dz, f.ux, f.uy, f.uz);
    float rdy = dot3(dx, dy, dz, f.wx, f.wy, f.wz);
    float rdz = dot3(dx, dy, dz, f.vx, f.vy, f.vz);
    float xx = rox * rdx, yy = roy * rdy, zz = roz * rdz;
    float n = (zz + yy) + xx;
    float sx = rox * rox, sy = roy * roy, sz = roz * roz;
    float m = (sz + sy) + sx;
    float rd_dxz = rdx * rdx + rdz * rdz;
    float ro_rd_dxz = xx + zz;
    float ro_dxz = sx + sz;
    float k = (m + f.R2) - f.r2;
    float four_n = 4.0f * n;
    float A = four_n;
    float B = (2.0f * k + four_n * n) - f.fourR2 * rd_dxz;
    float C = four_n * k - f.eightR2 * ro_rd_dxz;
    float D = k * k - f.fourR2 * ro_dxz;
    float t;
    bool found = quartic_nearest<DEBUG>(A, B, C, D, t, cnt);
    if (found && polish) found = torus_polish(rox, roy, roz, rdx, rdy, rdz, f.R, f.r, t);
    t_out = t;
    return found;
}

// ------------------------------------------------------------------ trace (fragment_shader.glsl:379-435)
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef RT_SCENE_SKIP
#define RT_SCENE_SKIP 1
#endif
#ifndef RT_MEMBER_PACKED
#define RT_MEMBER_PACKED 1      // member tests of the pooled trace with packed FP32 instructions; 0: one object per instruction
                                // (same-call A/B: equal on 64 tori and 256 spheres, packed 2.5 % faster on 1024 objects)
#endif
#ifndef RT_PAIR_PACKED
#define RT_PAIR_PACKED 0
#endif

// Centre of object `id` from the pair-interleaved bound records.
DEV void object_centre(const float *lds, int id, float &cx, float &cy, float &cz)
{
    const float *p = lds + 8 * (id >> 1) + (id & 1);
    cx = p[0]; cy = p[2]; cz = p[4];
}

// The exact intersection of one object (:74-92 sphere, :250-375 torus) given the terms of its
// bounding test; kd = the object's kind word (negative: sphere, else the index of its torus frame).
template <bool DEBUG>
DEV void candidate_exact(const float *lds, const SceneLayout &L, int polish, int kd, float a,
                         float ocx, float ocy, float ocz, float b, float c,
                         float dx, float dy, float dz, float &t, bool &hit, const Counters<DEBUG> &cnt)
{
    // [census: candidate_kind @ CAND]
    if (kd < 0) {
        // intersect_sphere (:74-92); c is the same dot(oc,oc) - r*r
        // [census: sphere @ SPHERE]
        CNT(EV_SPHERE_CALL);
        WCNT(WP_SPHERE); LCNT(WP_SPHERE_LANES);
        float b2 = 2.0f * b;
        float ds = b2 * b2 - (4.0f * a) * c;
        hit = false; t = 0.0f;
        if (!(ds < 0.0f)) {
            // [census: sphere_solve @ SPHERE_SOLVE]
            WCNT(WP_SPHERE_SOLVE);
            float sq = rt_sqrt(ds);
            float two_a = 2.0f * a;
            CNT(EV_SPHERE_ROOT);
            t = rt_div(-b2 - sq, two_a);
            // [census: sphere_root2 @ SPHERE_ROOT2]
            if (t < 0.0f) { CNT(EV_SPHERE_ROOT); WCNT(WP_SPHERE_ROOT2); t = rt_div(-b2 + sq, two_a); }
            // [census: sphere_end @ SPHERE_SOLVE]
            hit = t > 0.001f;
            if (hit) CNT(EV_SPHERE_HIT);
        }
    } else {
        // [census: torus_call @ TORUS]
        WCNT(WP_TORUS); LCNT(WP_TORUS_LANES);
        TorusFrame f = load_torus(lds, L.off_torus, kd);
        hit = torus_nearest<DEBUG>(f, ocx, ocy, ocz, dx, dy, dz, polish, t, cnt);
        if (hit) CNT(EV_TORUS_HIT);
    }
}

// Exact test of one object whose bounding test did not reject it (:405-425).
template <bool DEBUG>
DEV void test_candidate(const float *lds, const SceneLayout &L, int polish, int i, float a,
                        float ocx, float ocy, float ocz, float b, float c, float disc,
                        float dx, float dy, float dz, float &bt, int &bid, const Counters<DEBUG> &cnt)
{
    // [census: candidate @ CAND]
    CNT(EV_BOUND_TBOUND);
    WCNT(WP_CAND); LCNT(WP_CAND_LANES);
    const int kd = reinterpret_cast<const int *>(lds + L.off_kind)[i];
    // A sphere behind the ray (centre behind the origin, origin outside: the sphere a reflected or
    // shadow ray has just left, most of the time) cannot be hit: in intersect_sphere below the
    // discriminant is RN((2b)^2) minus a non-negative product, so its root is <= 2b and both
    // quotients are <= +0, which fails t > 0.001.  An object that is not hit changes nothing in
    // the reference's loop, so the instrumented flavour alone goes through the motions (its
    // counters are the reference's work).  b < 1e15 keeps (2b)^2 finite.
    if (!DEBUG && kd < 0 && b > 0.0f && b < 1e15f && c >= 0.0f) return;
    // t_bound > closest.t can only hold once something has been hit: while closest.t is
    // still 1e30 the comparison is false for every finite ray, so the sqrt is skipped.
    if (bt < 1e30f) {
        // [census: t_bound @ TBOUND]
        WCNT(WP_TBOUND);
        float t_bound = -b - rt_sqrt(disc);
        if (t_bound > bt) return;
    }
    // [census: candidate_kind @ CAND]
    float t; bool hit;
    candidate_exact<DEBUG>(lds, L, polish, kd, a, ocx, ocy, ocz, b, c, dx, dy, dz, t, hit, cnt);
    // [census: candidate_end @ CAND]
    if (hit && t < bt) { bt = t; bid = i; }
}

// intersect_plane (:107-123): y = -1, two-sided; tested after all objects (:428-433).
template <bool DEBUG>
DEV void offer_plane(float oy, float dy, int n, float &bt, int &bid, const Counters<DEBUG> &cnt)
{
    // [census: plane @ ITER]
    CNT(EV_PLANE_CALL);
    bool ok = __builtin_fabsf(dy) > 0.0001f;
    if (ok) {
        CNT(EV_PLANE_T);
        WCNT(WP_PLANE_DIV);
        float t = rt_div(-1.0f - oy, dy);
        if (t > 0.001f) {
            CNT(EV_PLANE_HIT);
            if (t < bt) { bt = t; bid = n; }
        }
    }
}

// The wave's mask of lanes in which `p` holds (the compare's own result, no detour through a 0/1
// value), and a lane's rank among the lanes of a mask that lie below it.
DEV unsigned long long wave_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DEV int rank_in_mask(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

// Closest hit as (t, id): id -1 none, 0..n-1 scene object, n ground plane.
// The bounding test (:399-402) runs on two objects per instruction with packed FP32 math
// (v_pk_add/v_pk_mul: each half is an ordinary correctly rounded binary32 operation, so
// the values are those of the scalar form); object order, and with it the tie rule "first
// index wins" (:423), is kept by resolving the pair's candidates in index order.
template <bool DEBUG>
DEV void trace_closest(const float *lds, const SceneLayout &L, int polish,
                       float ox, float oy, float oz, float dx, float dy, float dz,
                       float &best_t, int &best_id, const Counters<DEBUG> &cnt)
{
    // [census: trace_setup @ ITER]
    CNT(EV_TRACE);
    float bt = 1e30f;
    int bid = -1;
    const float4 *pairs = reinterpret_cast<const float4 *>(lds);
    const float a = dot3(dx, dy, dz, dx, dy, dz);    // intersect_sphere's dot(d,d) (:78)
    const int n = L.n;
    const int npairs = (n + 1) >> 1;
    // Rays that miss the sphere around the whole scene (L.hc, L.hR: it contains every object's
    // bounding sphere) skip the object loop: no object's binary32 bounding test can pass for
    // them, and an object whose test fails does nothing in the reference's loop.  The test is the
    // node test of trace_hierarchy (see there for the bound) with the scene sphere as the node, for
    // rays of near-unit direction (|a - 1| <= 1e-5, so sf^2 <= 2.5e-6 + 1.1e-5), and with
    // reach = |q| + hR replaced by the larger (|q|^2 + 1)/2 + hR, which saves the square root;
    // (reach + grow)^2 <= 1.01 reach^2.  Everything in it that depends on the scene alone is folded
    // into the three coefficients L.hq on the host (scene_test_polynomial, rounded up).  Its own
    // arithmetic is free to use fma: only its slack matters.
    // the first pair record is requested before the test: its LDS latency passes behind the test's
    // arithmetic (with no objects the read returns whatever follows in LDS and the loop does not run)
    float4 A = pairs[0], B = pairs[1];
    bool wave_skips = npairs <= 0;          // wave-uniform: no lane of the wave needs the object loop
    if (RT_SCENE_SKIP && !cnt.counting_events() && L.hR > 0.0f && n >= 3) {
        const float qx = ox - L.hc[0], qy = oy - L.hc[1], qz = oz - L.hc[2];
        const float q2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
        const float T = __builtin_fmaf(__builtin_fmaf(L.hq[2], q2, L.hq[1]), q2, L.hq[0]);
        const float bq = __builtin_fmaf(qz, dz, __builtin_fmaf(qy, dy, qx * dx));
        const float s = __builtin_fmaf(bq, bq, T - q2);
        // a lane skips if its direction is near unit length and s < 0 (a NaN keeps the loop); lane masks
        // straight from the two compares
        const unsigned long long skip = wave_mask(__builtin_fabsf(a - 1.0f) <= 1e-5f) & wave_mask(s < 0.0f);
        wave_skips = skip == wave_mask(true);
    }
#if RT_PAIR_PACKED
    const v2f OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
    const v2f DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
#endif
    if (wave_skips) { WCNT(WP_SCENE_SKIP); }
    else {
    for (int j = 0; j < npairs; ++j) {
        // [census: pair_test @ PAIR]
        WCNT(WP_PAIR);
#if RT_PAIR_PACKED
        v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, br2 = {B.z, B.w};
        v2f ocx = OX - cx, ocy = OY - cy, ocz = OZ - cz;
        v2f oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
        v2f c = oc2 - br2;
        // the record is consumed: request the next pair's (LDS broadcast read) into the same
        // registers; its latency is covered by the rest of this iteration
        const int jn = (j + 1 < npairs) ? j + 1 : j;
        A = pairs[2 * jn]; B = pairs[2 * jn + 1];
        v2f b = (ocz * DZ + ocy * DY) + ocx * DX;
        v2f disc = b * b - c;
#else
        // One object per instruction: a packed FP32 instruction takes a SIMD as long as the two
        // scalar ones it replaces, and the scalar form needs no duplicated ray operands.
        v2f ocx, ocy, ocz, b, c, disc;
        ocx.x = ox - A.x; ocx.y = ox - A.y; ocy.x = oy - A.z; ocy.y = oy - A.w;
        ocz.x = oz - B.x; ocz.y = oz - B.y;
        c.x = ((ocz.x * ocz.x + ocy.x * ocy.x) + ocx.x * ocx.x) - B.z;
        c.y = ((ocz.y * ocz.y + ocy.y * ocy.y) + ocx.y * ocx.y) - B.w;
        const int jn = (j + 1 < npairs) ? j + 1 : j;
        A = pairs[2 * jn]; B = pairs[2 * jn + 1];
        b.x = (ocz.x * dz + ocy.x * dy) + ocx.x * dx;
        b.y = (ocz.y * dz + ocy.y * dy) + ocx.y * dx;
        disc.x = b.x * b.x - c.x;
        disc.y = b.y * b.y - c.y;
#endif
        if (DEBUG) {
            const int *kind = reinterpret_cast<const int *>(lds + L.off_kind);
            if (kind[2 * j] >= 0) CNT(EV_BOUND_TORUS_RADIUS);
            if (2 * j + 1 < n && kind[2 * j + 1] >= 0) CNT(EV_BOUND_TORUS_RADIUS);
            if (disc.x < 0.0f) CNT(EV_BOUND_CULL_DISC);
            if (2 * j + 1 < n && disc.y < 0.0f) CNT(EV_BOUND_CULL_DISC);
        }
        if (!(disc.x < 0.0f))
            test_candidate<DEBUG>(lds, L, polish, 2 * j, a, ocx.x, ocy.x, ocz.x, b.x, c.x, disc.x, dx, dy, dz, bt, bid, cnt);
        // an odd object count is padded with a record whose radius^2 is -inf: disc = -inf
        if (!(disc.y < 0.0f))
            test_candidate<DEBUG>(lds, L, polish, 2 * j + 1, a, ocx.y, ocy.y, ocz.y, b.y, c.y, disc.y, dx, dy, dz, bt, bid, cnt);
    }
    }
    // [census: trace_end @ ITER]
    offer_plane<DEBUG>(oy, dy, n, bt, bid, cnt);
    best_t = bt; best_id = bid;
}

// ------------------------------------------------------------------ trace through the sphere hierarchy
// [census: hierarchy @ ITER]
//
// The reference's loop (:386-426) only ever acts on objects whose bounding test passes
// (disc >= 0); for those, what it does depends on the order (the t_bound > closest.t skip and
// "first index wins").  trace_hierarchy therefore (1) finds exactly the set of objects whose
// bounding test passes, by walking two levels of enclosing spheres and running the reference's
// own bounding arithmetic on the members of the leaf nodes it reaches, and (2) replays the
// reference's loop body over that set in ascending object index.  Step 2 is the reference
// algorithm with the iterations it would have culled left out, so the result is the same
// (t, id) for every ray, whatever the enclosing spheres look like -- provided step 1 never
// drops an object that passes.  A node (centre g, radius R) is kept when
//     disc_node + slack >= 0     (or the comparison is unordered: NaN/inf keep everything)
// where disc_node is the bounding discriminant of the sphere (g, R + grow), with
//     reach = |o - scene centre| + scene radius  (>= |oc| + radius for every object and node),
//     grow  = sf * reach,  sf^2 = 2.5e-6 + 1.1 |dot(d,d) - 1|,
//     slack = sf^2 (reach^2 + (reach + grow)^2).
// Why that is enough: a member whose binary32 disc is >= 0 has its centre within
// br + sqrt(E) of the ray's line, where E <= (17u + |dot(d,d) - 1|) |oc|^2 + 2u br^2
// (u = 2^-24) bounds the rounding of the member test plus the direction's deviation from unit
// length, so sqrt(E) <= grow; the member's sphere lies inside the node's, so the line passes
// within R + grow of g, i.e. the exact node discriminant is >= 0, and the binary32 one
// differs from it by less than sf^2 (|oc_node|^2 + (R + grow)^2) <= slack.
//
//
// Spheres behind the ray.  A sphere whose bounding test passes with b = dot(oc, d) > 0 and
// c = dot(oc, oc) - r^2 >= 0 (origin outside, centre behind) cannot be hit: in intersect_sphere
// (:74-92) the discriminant is RN((2b)^2) minus a non-negative product, so its root is <= 2b and
// both quotients (-2b -+ sqrt) / 2a are <= +0, which fails t > 0.001.  An object that is not hit
// changes nothing in the reference's loop, so such spheres are left out of the candidates (exact;
// b < 1e15 keeps (2b)^2 finite).  Tori are never left out: the binary32 quartic solve can return
// roots that geometry rules out.  A node whose members are all spheres (radius stored negated)
// is skipped as a whole when its sphere, grown as above, lies behind the origin:
// b_node > (R + grow) max(1, a).  Then every member has b > r + 0.99 grow > 0 and
// c > 0.98 grow^2 >= 2.4e-6 reach^2, far above the rounding of either (<= 6e-7 reach^2), so each
// of them would have been left out individually.  Used only for 0.5 < a < 2 and reach < 1e12.
//
// The candidates are queued per ray in LDS as 16-bit object indices; a ray that collects
// more than RT_QUEUE of them walks all objects in step 2 instead (same result, slower).
#ifndef RT_SCAN_LEAVES
#define RT_SCAN_LEAVES 32       // hierarchies of at most this many leaf nodes are not walked from the top (see hierarchy_kept_leaves)
#endif
static_assert(RT_SCAN_LEAVES <= 32, "the scan keeps its result in one 32-bit mask");
// [census: node_keep @ 0]
DEV void node_keep(float4 A, float4 B, v2f OX, v2f OY, v2f OZ, v2f DX, v2f DY, v2f DZ, float grow, float slack,
                   float behind_scale, bool &keep0, bool &keep1)
{
    v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y};
    v2f R = {__builtin_fabsf(B.z), __builtin_fabsf(B.w)};
    v2f ocx = OX - cx, ocy = OY - cy, ocz = OZ - cz;
    // The node test only has to be conservative (its slack is 25 times the rounding of any evaluation
    // order), so its products and sums are fused: 14 packed instructions instead of 20.
    v2f oc2 = __builtin_elementwise_fma(ocz, ocz, __builtin_elementwise_fma(ocy, ocy, ocx * ocx));
    v2f G = {grow, grow}, S = {slack, slack};
    v2f Re = R + G;
    v2f b = __builtin_elementwise_fma(ocz, DZ, __builtin_elementwise_fma(ocy, DY, ocx * DX));
    v2f s = __builtin_elementwise_fma(b, b, S - __builtin_elementwise_fma(-Re, Re, oc2));     // b^2 - (oc2 - Re^2) + slack
    // behind_scale = max(1, a) for rays that may use the shortcut, +inf otherwise
    v2f lim = Re * (v2f){behind_scale, behind_scale};
    const bool behind0 = __builtin_signbitf(B.z) && b.x > lim.x;
    const bool behind1 = __builtin_signbitf(B.w) && b.y > lim.y;
    keep0 = !(s.x < 0.0f) && !behind0;
    keep1 = !(s.y < 0.0f) && !behind1;
}

// Growth, slack and behind-limit of the node test for one ray (see the comment above node_keep).
// [census: node_constants @ ITER]
struct NodeTest { float grow, slack, behind_scale; };
DEV NodeTest node_test_constants(const SceneLayout &L, float ox, float oy, float oz, float a)
{
    // every object and node lies inside the scene sphere (L.hc, L.hR), so |oc| + R <= reach = |o - hc| + hR for all of them
    float sf2 = 2.5e-6f + 1.1f * __builtin_fabsf(a - 1.0f);
    const float sf = __builtin_amdgcn_sqrtf(sf2) * 1.02f;
    const float qx = ox - L.hc[0], qy = oy - L.hc[1], qz = oz - L.hc[2];
    const float reach = (__builtin_amdgcn_sqrtf((qz * qz + qy * qy) + qx * qx) + L.hR) * 1.0001f;
    NodeTest nt;
    nt.grow = sf * reach;
    nt.slack = (sf2 * 1.05f) * (reach * reach + (reach + nt.grow) * (reach + nt.grow)) + 1e-30f;
    // all-sphere nodes behind the origin are skipped, for ordinary rays only (else the limit is +inf)
    const bool shortcut_ok = a > 0.5f && a < 2.0f && reach < 1e12f;
    nt.behind_scale = shortcut_ok ? (a > 1.0f ? a : 1.0f) * 1.0001f : __builtin_inff();
    return nt;
}
// [census: top_nodes @ ITER]
// bit j: top node j is kept.  hl = the hierarchy sections in LDS, starting at the top nodes.
DEV unsigned int kept_top_nodes(const float *hl, const SceneLayout &L, const NodeTest &nt,
                                float ox, float oy, float oz, float dx, float dy, float dz)
{
    const v2f OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
    const v2f DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
    const float4 *top = reinterpret_cast<const float4 *>(hl);
    const int n2 = L.n2;
    unsigned int top_mask = 0;
    for (int j = 0; j < ((n2 + 1) >> 1); ++j) {
        bool k0, k1;
        node_keep(top[2 * j], top[2 * j + 1], OX, OY, OZ, DX, DY, DZ, nt.grow, nt.slack, nt.behind_scale, k0, k1);
        top_mask |= (k0 ? 1u : 0u) << (2 * j);
        top_mask |= (k1 ? 2u : 0u) << (2 * j);
    }
    return top_mask & (n2 >= 32 ? ~0u : (1u << n2) - 1u);
}
// [census: leaves_under_top @ TOP]
// bit k: leaf node tn * RT_FAN + k (one of the RT_FAN leaf nodes under top node tn) is kept
DEV unsigned int kept_leaves_under(const float *hl, const SceneLayout &L, const NodeTest &nt, int tn,
                                   float ox, float oy, float oz, float dx, float dy, float dz)
{
    const v2f OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
    const v2f DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
    const float4 *leafn = reinterpret_cast<const float4 *>(hl + (L.off_leaf - L.off_top));
    const int n1 = L.n1;
    const int last_leaf_pair = (n1 - 1) >> 1;
    const int first = tn * RT_FAN;                       // first leaf node of this top node
    const int nleaf = (n1 - first < RT_FAN) ? n1 - first : RT_FAN;
    unsigned int bits = 0;
#pragma unroll
    for (int kk = 0; kk < RT_FAN / 2; ++kk) {
        int lp = (first >> 1) + kk;
        lp = lp < last_leaf_pair ? lp : last_leaf_pair;   // stay inside the section; masked below
        bool k0, k1;
        node_keep(leafn[2 * lp], leafn[2 * lp + 1], OX, OY, OZ, DX, DY, DZ, nt.grow, nt.slack, nt.behind_scale, k0, k1);
        bits |= (k0 ? 1u : 0u) << (2 * kk);
        bits |= (k1 ? 2u : 0u) << (2 * kk);
    }
    return bits & ((1u << nleaf) - 1u);     // the last top node may have fewer than RT_FAN leaf nodes
}

// Step 1 down to the leaf level: bit masks of the leaf nodes this lane's ray keeps (0..63 / 64..127).
// The top nodes are tested by all lanes together; below them every lane walks its own list of
// kept nodes (bit masks), so a wave of incoherent rays pays for the longest list of any lane,
// not for the union of all lists.
template <bool DEBUG>
DEV void hierarchy_kept_leaves(const float *hl, const SceneLayout &L,
                               float ox, float oy, float oz, float dx, float dy, float dz,
                               unsigned long long &leaf_lo_out, unsigned long long &leaf_hi_out, const Counters<DEBUG> &cnt)
{
    const float a = dot3(dx, dy, dz, dx, dy, dz);    // intersect_sphere's dot(d,d) (:78)
    const NodeTest nt = node_test_constants(L, ox, oy, oz, a);
    unsigned long long leaf_lo = 0, leaf_hi = 0;        // kept leaf nodes 0..63 / 64..127
    if (L.n1 <= RT_SCAN_LEAVES) {
        // Few leaf nodes: every lane tests all of them, the wave in step (uniform addresses: LDS
        // broadcasts, no loop of its own per lane).  The two-level walk below issues as many
        // iterations as the lane with the most kept top nodes needs, each with its own index
        // arithmetic.  Same-call A/B at 4K: 48 / 96 objects depth 4 -7 / -5 %, 64 tori and 256
        // spheres -1.5 %, 256 mixed objects +2 %; with 64 leaf nodes (512 objects) +11 %.
        const float4 *leafn = reinterpret_cast<const float4 *>(hl + (L.off_leaf - L.off_top));
        const v2f OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
        const v2f DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
        unsigned int kept = 0;
        const int npairs = (L.n1 + 1) >> 1;
        for (int lp = 0; lp < npairs; ++lp) {
            HSTAT(HS_TOP);
            bool k0, k1;
            node_keep(leafn[2 * lp], leafn[2 * lp + 1], OX, OY, OZ, DX, DY, DZ, nt.grow, nt.slack, nt.behind_scale, k0, k1);
            kept |= ((k0 ? 1u : 0u) | (k1 ? 2u : 0u)) << (2 * lp);
        }
        kept &= L.n1 >= 32 ? ~0u : (1u << L.n1) - 1u;
        leaf_lo_out = kept; leaf_hi_out = 0ull;
        return;
    }
    unsigned int top_mask = kept_top_nodes(hl, L, nt, ox, oy, oz, dx, dy, dz);
    while (top_mask) {
        HSTAT(HS_TOP);
        const int tn = __builtin_ctz(top_mask);
        top_mask &= top_mask - 1;
        const unsigned int bits = kept_leaves_under(hl, L, nt, tn, ox, oy, oz, dx, dy, dz);
        const int first = tn * RT_FAN;
        const unsigned long long placed = (unsigned long long)bits << (first & 63);
        if (first < 64) leaf_lo |= placed; else leaf_hi |= placed;
    }
    leaf_lo_out = leaf_lo; leaf_hi_out = leaf_hi;
}

// queue: this lane's column of 16-bit entries, QSTRIDE entries apart (the threads of the workgroup, or the lanes of a wave)
template <bool DEBUG, int QSTRIDE = 256>
DEV void trace_hierarchy(const float *sc, const float *hl, const SceneLayout &L, int polish, unsigned short *queue,
                         float ox, float oy, float oz, float dx, float dy, float dz,
                         float &best_t, int &best_id, const Counters<DEBUG> &cnt)
{
    const float a = dot3(dx, dy, dz, dx, dy, dz);    // intersect_sphere's dot(d,d) (:78)
    const int n = L.n;
    const v2f OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
    const v2f DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
    const float4 *memb = reinterpret_cast<const float4 *>(hl + (L.off_member - L.off_top));
    const unsigned short *slot_id = reinterpret_cast<const unsigned short *>(hl + (L.off_slot_id - L.off_top));
    unsigned long long leaf_lo, leaf_hi;
    hierarchy_kept_leaves<DEBUG>(hl, L, ox, oy, oz, dx, dy, dz, leaf_lo, leaf_hi, cnt);
    int count = 0;
    while (leaf_lo | leaf_hi) {
        HSTAT(HS_LEAF);
        int leaf;
        if (leaf_lo) { leaf = __builtin_ctzll(leaf_lo); leaf_lo &= leaf_lo - 1; }
        else { leaf = 64 + __builtin_ctzll(leaf_hi); leaf_hi &= leaf_hi - 1; }
#pragma unroll
        for (int m = 0; m < RT_LEAF / 2; ++m) {
            const int mp = leaf * (RT_LEAF / 2) + m;
            const float4 A = memb[2 * mp], B = memb[2 * mp + 1];
            // the reference's bounding test (:399-402), operation for operation
            v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, br2 = {B.z, B.w};
            v2f ocx = OX - cx, ocy = OY - cy, ocz = OZ - cz;
            v2f oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
            v2f c = oc2 - br2;
            v2f b = (ocz * DZ + ocy * DY) + ocx * DX;
            v2f disc = b * b - c;
            // (spheres that lie behind the ray are left out, see above)
            if (!(disc.x < 0.0f)) {
                const unsigned int sid = slot_id[2 * mp];
                if (!(!(sid & 0x8000u) && b.x > 0.0f && b.x < 1e15f && c.x >= 0.0f)) {
                    if (count < RT_QUEUE) queue[count * QSTRIDE] = (unsigned short)(sid & 0x7fffu);
                    count++;
                }
            }
            if (!(disc.y < 0.0f)) {
                const unsigned int sid = slot_id[2 * mp + 1];
                if (!(!(sid & 0x8000u) && b.y > 0.0f && b.y < 1e15f && c.y >= 0.0f)) {
                    if (count < RT_QUEUE) queue[count * QSTRIDE] = (unsigned short)(sid & 0x7fffu);
                    count++;
                }
            }
        }
    }

    // ---- step 2: the reference's loop body over the candidates in ascending index
    CNT(EV_TRACE);
    float bt = 1e30f;
    int bid = -1;
    const bool walk_all = count > RT_QUEUE;
    int last = -1;
    for (;;) {
        int i;
        if (!walk_all) {
            int mnext = 0x7fffffff;
            for (int q = 0; q < count; ++q) {
                HSTAT(HS_SCAN);
                int v = queue[q * QSTRIDE];
                mnext = (v > last && v < mnext) ? v : mnext;
            }
            if (mnext == 0x7fffffff) break;
            i = mnext;
        } else {
            i = last + 1;
            if (i >= n) break;
        }
        last = i;
        HSTAT(HS_CAND);
        const float *rec = sc + 8 * (i >> 1) + (i & 1);
        float ocx = ox - rec[0], ocy = oy - rec[2], ocz = oz - rec[4];
        float oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
        float c = oc2 - rec[6];
        float b = (ocz * dz + ocy * dy) + ocx * dx;
        float disc = b * b - c;
        if (disc < 0.0f) continue;
        test_candidate<DEBUG>(sc, L, polish, i, a, ocx, ocy, ocz, b, c, disc, dx, dy, dz, bt, bid, cnt);
    }
    offer_plane<DEBUG>(oy, dy, n, bt, bid, cnt);
    best_t = bt; best_id = bid;
}

// Inclusive prefix sum over the 64 lanes in six VALU instructions with DPP operands (shifts inside
// the rows of 16 lanes, then lane 15 / 31 broadcast into the following rows); lanes without a
// source add 0.
DEV int wave_inclusive_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return x;
}

// ------------------------------------------------------------------ the same trace, the wave's work pooled
//
// trace_hierarchy above lets every lane walk its own lists, so the wave issues as many member tests
// and exact intersections as its *longest* list has entries (measured on 256 spheres at depth 16:
// 7.7 leaf iterations per trace with 25 lanes in them, 4.2 candidate iterations with 16).  Here the
// lists of all 64 rays are pooled in LDS and dealt out evenly:
//   1. every lane finds the leaf nodes its ray keeps (hierarchy_kept_leaves, as before) and writes
//      them as (ray, leaf) items; lane j of a round takes item j, reads that ray from an LDS table
//      and runs the reference's bounding test (:399-402) on the leaf's members;
//   2. objects that pass (minus the spheres behind the ray, see trace_hierarchy) become
//      (ray, object) items, dealt out the same way for the exact intersection (candidate_exact: the
//      same binary32 operations on the same operands as in the per-lane form);
//   3. per ray, LDS atomics keep the two smallest (t, object index) keys over its hits with
//      t < 1e30 (t > 0.001 is positive, so the bit pattern orders like the number; equal t orders
//      by index).
// What the reference's loop (:386-426) makes of the hits: it visits them in ascending index, skips
// object j when t_bound(j) > closest.t, and takes j when t(j) < closest.t.  Let m be the smallest
// key.  Hits visited before m have t > t(m), hits after it t >= t(m): if m is not skipped it is
// taken and never replaced.  m is skipped only if a hit taken before it has t < t_bound(m), so if no
// other hit at all has t < t_bound(m) -- the second smallest key decides that -- the result is m.
// Otherwise (two hits within rounding of each other, or a torus root that geometry rules out) the
// lane runs the reference's loop over all objects.  All 64 lanes must enter; `mine` marks the
// lanes that hold a ray.  wl = this wave's RT_FLAT_DWORDS dwords of LDS.
#define RT_FLAT_QUOTA 16                           // leaf items a lane contributes per chunk
#define RT_FLAT_ITEMS (64 * RT_FLAT_QUOTA)
#define RT_FLAT_CAND 640                            // candidate items: < 64 + < 64 left over (mixed stack, tori stack) + up to 64 x RT_LEAF new ones per round
// ray table: origin, direction; with the pooled top-node loop also grow, slack and behind limit of the node test
#define RT_FLAT_RAY_FIELDS(toppool) ((toppool) ? 9 : 6)
#define RT_FLAT_DWORDS(toppool) (RT_FLAT_RAY_FIELDS(toppool) * 64 + 4 * 64 + RT_FLAT_ITEMS / 2 + RT_FLAT_CAND)
static_assert(RT_FLAT_DWORDS(false) * 2 >= RT_QUEUE * 64, "the wave's LDS holds the candidate queues of a replay");
static_assert(RT_FLAT_CAND >= 2 * 64 + 64 * RT_LEAF, "one round of member tests must fit next to a partial round of candidates of either kind");

// WALK_REPLAY: the rare replay (see above) goes through the per-lane walk (trace_hierarchy) instead of over all
// objects.  Same-call A/B: 1024 objects 8K 16 spp 998 -> 971 ms; the single-sample kernels, which run at 128
// registers, lose 5 % with the walk inlined next to the pooled trace, so only the multi-sample ones carry it.
template <bool DEBUG, bool TOPPOOL, bool WALK_REPLAY = false>
DEV void trace_hierarchy_flat(const float *sc, const float *hl, const SceneLayout &L, int polish, float *wl, bool mine,
                              float ox, float oy, float oz, float dx, float dy, float dz,
                              float &best_t, int &best_id, const Counters<DEBUG> &cnt)
{
    // [census: flat_setup @ ITER]
    const int lane = (int)(threadIdx.x & 63u);
    const int n = L.n;
    constexpr int RF = RT_FLAT_RAY_FIELDS(TOPPOOL);
    float *ray = wl;                                                                // [RF][64]
    unsigned long long *key1 = reinterpret_cast<unsigned long long *>(wl + RF * 64);  // [64] smallest key per ray
    unsigned long long *key2 = key1 + 64;                                           // [64] second smallest
    unsigned short *items = reinterpret_cast<unsigned short *>(wl + (RF + 4) * 64);   // (ray << 8) | leaf (or top node)
    unsigned int *clist = reinterpret_cast<unsigned int *>(wl + (RF + 4) * 64 + RT_FLAT_ITEMS / 2);   // (ray << 16) | object
    const float4 *memb = reinterpret_cast<const float4 *>(hl + (L.off_member - L.off_top));
    const unsigned short *slot_id = reinterpret_cast<const unsigned short *>(hl + (L.off_slot_id - L.off_top));
    const int *kind = reinterpret_cast<const int *>(sc + L.off_kind);
    constexpr unsigned long long NOKEY = ~0ull;

    ray[lane] = ox; ray[64 + lane] = oy; ray[128 + lane] = oz;
    ray[192 + lane] = dx; ray[256 + lane] = dy; ray[320 + lane] = dz;
    key1[lane] = NOKEY; key2[lane] = NOKEY;
    unsigned long long leaf_lo = 0, leaf_hi = 0;
    if constexpr (!TOPPOOL) {
        if (mine) hierarchy_kept_leaves<DEBUG>(hl, L, ox, oy, oz, dx, dy, dz, leaf_lo, leaf_hi, cnt);
    } else {
        // Large hierarchies (the host's choice: more than 64 leaf nodes): the leaf nodes under the
        // kept top nodes are pooled like the rest, as (ray, top node) items -- a lane of its own
        // runs 7 iterations with 30 lanes in them on 1024 objects; the kept leaves are OR-ed into
        // the ray's two mask words, which live in the candidate list's memory (unused until the leaf
        // rounds).  Same-call A/B: 1024 objects 8K 16 spp 1169 -> 1117 ms, 1080p -5 %; 512 objects
        // (64 leaf nodes) +11 %; and it costs the kernel 10 spilled dwords, which is why kernels
        // for smaller scenes do not carry it.
        unsigned long long *lmask = reinterpret_cast<unsigned long long *>(clist);      // [64][2]
        unsigned int top_mask = 0;
        NodeTest nt = {0.0f, 0.0f, 0.0f};
        if (mine) {
            nt = node_test_constants(L, ox, oy, oz, dot3(dx, dy, dz, dx, dy, dz));
            top_mask = kept_top_nodes(hl, L, nt, ox, oy, oz, dx, dy, dz);
        }
        ray[384 + lane] = nt.grow; ray[448 + lane] = nt.slack; ray[512 + lane] = nt.behind_scale;
        lmask[2 * lane] = 0ull; lmask[2 * lane + 1] = 0ull;
        for (;;) {
            int c = __popc(top_mask);
            c = c < RT_FLAT_QUOTA ? c : RT_FLAT_QUOTA;
            const int incl = wave_inclusive_sum(c);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total == 0) break;
            int at = incl - c;
            for (int k = 0; k < c; ++k) {
                const int tn = __builtin_ctz(top_mask);
                top_mask &= top_mask - 1;
                items[at++] = (unsigned short)((lane << 8) | tn);
            }
            for (int base = 0; base < total; base += 64) {
                HSTAT(HS_TOP);
                const int j = base + lane;
                if (j < total) {
                    const unsigned int it = items[j];
                    const int r = (int)(it >> 8), tn = (int)(it & 0xffu);
                    const NodeTest rnt = {ray[384 + r], ray[448 + r], ray[512 + r]};
                    const unsigned int bits = kept_leaves_under(hl, L, rnt, tn, ray[r], ray[64 + r], ray[128 + r],
                                                                ray[192 + r], ray[256 + r], ray[320 + r]);
                    const int first = tn * RT_FAN;
                    if (bits) atomicOr(&lmask[2 * r + (first >> 6)], (unsigned long long)bits << (first & 63));
                }
            }
        }
        leaf_lo = lmask[2 * lane]; leaf_hi = lmask[2 * lane + 1];
    }

    // Candidate items.  clist is one array used from both ends: the items the leaf rounds append
    // (spheres and tori mixed) are a stack that grows upwards from clist[0]; the exact rounds take
    // them off its top 64 at a time, intersect the spheres among them at once and move the tori to
    // a second stack that grows downwards from clist[RT_FLAT_CAND - 1], so that the quartic solve
    // (about seven times the instructions of the sphere's quadratic) runs in rounds of its own with
    // all lanes on tori instead of in every round with the few lanes that happen to hold one.  An
    // item moved to the second stack has just left the first, so both always fit: fewer than 64 of
    // each wait when a leaf round starts, which appends at most 64 x RT_LEAF.
    cnt.tick(HS_T_SCAN);
    int ncand = 0, ntor = 0;          // wave-uniform: items on the mixed stack / on the tori stack
    auto offer_hit = [&](int r, int i, float t, bool hit) {
        if (hit && t < 1e30f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned int)i;
            const unsigned long long old = atomicMin(&key1[r], key);
            const unsigned long long loser = old > key ? old : key;
            if (loser != NOKEY) atomicMin(&key2[r], loser);
        }
    };
    // bounding-test terms of object i for ray r, as the reference computes them (:399-402), and the exact intersection
    auto exact_item = [&](unsigned int it, bool torus) {
        const int r = (int)(it >> 16), i = (int)(it & 0x7fffu);
        const float rox = ray[r], roy = ray[64 + r], roz = ray[128 + r];
        const float rdx = ray[192 + r], rdy = ray[256 + r], rdz = ray[320 + r];
        const float a = dot3(rdx, rdy, rdz, rdx, rdy, rdz);
        const float *rec = sc + 8 * (i >> 1) + (i & 1);
        const float ocx = rox - rec[0], ocy = roy - rec[2], ocz = roz - rec[4];
        const float oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
        const float c = oc2 - rec[6];
        const float b = (ocz * rdz + ocy * rdy) + ocx * rdx;
        float t; bool hit;
        int kd = -1;
        if (torus) { kd = kind[i]; kd = kd < 0 ? 0 : kd; }
        candidate_exact<DEBUG>(sc, L, polish, torus ? kd : -1, a, ocx, ocy, ocz, b, c, rdx, rdy, rdz, t, hit, cnt);
        offer_hit(r, i, t, hit);
    };
    // full rounds while 64 items wait; with `all` also the partial last ones
    auto exact_rounds = [&](bool all) {
        cnt.tick(HS_T_LEAF);
        while (ncand >= 64 || (all && ncand > 0)) {
            // [census: flat_exact_round @ FLAT_EXACT]
            HSTAT(HS_CAND);
            const int first = ncand >= 64 ? ncand - 64 : 0;
            const int j = first + lane;
            const bool have = j < ncand;
            const unsigned int it = clist[have ? j : 0];
            const bool is_torus = have && (it & 0x8000u);
            const unsigned long long mt = wave_mask(is_torus);
            if (mt != 0ull) {
                if (is_torus) clist[RT_FLAT_CAND - 1 - (ntor + rank_in_mask(mt))] = it;
                ntor += __popcll(mt);
            }
            if (have && !is_torus) exact_item(it, false);
            ncand = first;
        }
        while (ntor >= 64 || (all && ntor > 0)) {
            // [census: flat_torus_round @ FLAT_TORUS]
            HSTAT(HS_TORUS_ROUND);
            const int take = ntor >= 64 ? 64 : ntor;
            if (lane < take) exact_item(clist[RT_FLAT_CAND - ntor + lane], true);
            ntor -= take;
        }
        cnt.tick(HS_T_EXACT);
    };

    const unsigned int *slot_pair = reinterpret_cast<const unsigned int *>(slot_id);     // the ids of slots 2k, 2k + 1
    for (;;) {
        // this chunk's items: up to RT_FLAT_QUOTA kept leaves per lane, at the lane's offset in the wave's list
        // [census: flat_items @ ITER]
        int c = __popcll(leaf_lo) + __popcll(leaf_hi);
        c = c < RT_FLAT_QUOTA ? c : RT_FLAT_QUOTA;
        const int incl = wave_inclusive_sum(c);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) break;
        int at = incl - c;
        for (int k = 0; k < c; ++k) {
            int leaf;
            if (leaf_lo) { leaf = __builtin_ctzll(leaf_lo); leaf_lo &= leaf_lo - 1; }
            else { leaf = 64 + __builtin_ctzll(leaf_hi); leaf_hi &= leaf_hi - 1; }
            items[at++] = (unsigned short)((lane << 8) | leaf);
        }
        cnt.tick(HS_T_SCAN);
        for (int base = 0; base < total; base += 64) {
            // [census: flat_leaf_round @ FLAT_LEAF]
            HSTAT(HS_LEAF);
            const int j = base + lane;
            const bool have = j < total;
            // a lane without an item tests the leaf of empty slots that follows the last leaf node: nothing passes
            const unsigned int it = have ? (unsigned int)items[j] : (unsigned int)((lane << 8) | L.n1);
            const int r = (int)(it >> 8), leaf = (int)(it & 0xffu);
            const float rox = ray[r], roy = ray[64 + r], roz = ray[128 + r];
            const float rdx = ray[192 + r], rdy = ray[256 + r], rdz = ray[320 + r];
            const unsigned int entry_hi = (unsigned int)r << 16;
#pragma unroll
            for (int m = 0; m < RT_LEAF / 2; ++m) {
                const int mp = leaf * (RT_LEAF / 2) + m;
                const float4 A = memb[2 * mp], B = memb[2 * mp + 1];
                // the reference's bounding test (:399-402), operation for operation
#if RT_MEMBER_PACKED
                const v2f OX = {rox, rox}, OY = {roy, roy}, OZ = {roz, roz};
                const v2f DX = {rdx, rdx}, DY = {rdy, rdy}, DZ = {rdz, rdz};
                v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, br2 = {B.z, B.w};
                v2f ocx = OX - cx, ocy = OY - cy, ocz = OZ - cz;
                v2f oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
                v2f cc = oc2 - br2;
                v2f b = (ocz * DZ + ocy * DY) + ocx * DX;
                v2f disc = b * b - cc;
#else
                v2f ocx, ocy, ocz, cc, b, disc;
                ocx.x = rox - A.x; ocx.y = rox - A.y; ocy.x = roy - A.z; ocy.y = roy - A.w;
                ocz.x = roz - B.x; ocz.y = roz - B.y;
                cc.x = ((ocz.x * ocz.x + ocy.x * ocy.x) + ocx.x * ocx.x) - B.z;
                cc.y = ((ocz.y * ocz.y + ocy.y * ocy.y) + ocx.y * ocx.y) - B.w;
                b.x = (ocz.x * rdz + ocy.x * rdy) + ocx.x * rdx;
                b.y = (ocz.y * rdz + ocy.y * rdy) + ocx.y * rdx;
                disc.x = b.x * b.x - cc.x;
                disc.y = b.y * b.y - cc.y;
#endif
                // A sphere behind the ray (b > 0, c >= 0, see trace_hierarchy) is left out.  In one word: the
                // sign bits of b and of c and the torus flag of the slot id (bit 15 of each half) or-ed; the
                // member can be a candidate only if the result is negative.  That also leaves out spheres with
                // b = +0 (then disc >= 0 needs c <= 0, and c = +0 gives both roots 0: no hit), with b >= 1e15
                // (the only hit such a sphere can report is t = inf, which never beats 1e30) and with a NaN of
                // positive sign in b or c (the sphere's solve then yields NaN: no hit).
                const unsigned int sp = slot_pair[mp];
                const unsigned int k0 = __float_as_uint(b.x) | __float_as_uint(cc.x) | (sp << 16);
                const unsigned int k1 = __float_as_uint(b.y) | __float_as_uint(cc.y) | sp;
                const bool p0 = !(disc.x < 0.0f) && (int)k0 < 0;
                const bool p1 = !(disc.y < 0.0f) && (int)k1 < 0;
                // (each mask straight from its compare: a mask of the combined condition would go through a 0/1 value)
                const unsigned long long m0 = wave_mask(!(disc.x < 0.0f)) & wave_mask((int)k0 < 0);
                const unsigned long long m1 = wave_mask(!(disc.y < 0.0f)) & wave_mask((int)k1 < 0);
                if (p0) clist[ncand + rank_in_mask(m0)] = entry_hi | (sp & 0xffffu);
                ncand += __popcll(m0);
                if (p1) clist[ncand + rank_in_mask(m1)] = entry_hi | (sp >> 16);
                ncand += __popcll(m1);
            }
            if (ncand + ntor > RT_FLAT_CAND - 64 * RT_LEAF) exact_rounds(false);
        }
    }
    cnt.tick(HS_T_LEAF);
    exact_rounds(true);

    // [census: flat_owner @ ITER]
    float bt = 1e30f;
    int bid = -1;
    bool replay = false;
    if (mine) {
        const unsigned long long k1 = key1[lane];
        if (k1 != NOKEY) {
            const unsigned long long k2 = key2[lane];
            bid = (int)(unsigned int)k1;
            bt = __uint_as_float((unsigned int)(k1 >> 32));
            if (k2 != NOKEY) {
                // t_bound of the winner (:412): another hit in front of it could have made the reference skip it
                const float *rec = sc + 8 * (bid >> 1) + (bid & 1);
                const float ocx = ox - rec[0], ocy = oy - rec[2], ocz = oz - rec[4];
                const float oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
                const float c = oc2 - rec[6];
                const float b = (ocz * dz + ocy * dy) + ocx * dx;
                const float t_bound = -b - rt_sqrt(b * b - c);
                replay = __uint_as_float((unsigned int)(k2 >> 32)) < t_bound;
            }
        }
    }
    if (wave_mask(replay) != 0ull) {
        // [census: flat_replay @ 0]
        HSTAT(HS_SCAN);
        if (replay) {
            if constexpr (WALK_REPLAY) {
                // the reference's loop over the objects whose bounding test passes, in ascending index, found
                // once more by this lane alone through the hierarchy (the wave's lists are spent by now;
                // their LDS serves as the lane's candidate queue); ends with the ground plane like this function
                trace_hierarchy<DEBUG, 64>(sc, hl, L, polish, reinterpret_cast<unsigned short *>(wl) + lane,
                                           ox, oy, oz, dx, dy, dz, best_t, best_id, cnt);
            } else {
                // the reference's loop as it stands
                const float a = dot3(dx, dy, dz, dx, dy, dz);
                bt = 1e30f; bid = -1;
                for (int i = 0; i < n; ++i) {
                    const float *rec = sc + 8 * (i >> 1) + (i & 1);
                    const float ocx = ox - rec[0], ocy = oy - rec[2], ocz = oz - rec[4];
                    const float oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
                    const float c = oc2 - rec[6];
                    const float b = (ocz * dz + ocy * dy) + ocx * dx;
                    const float disc = b * b - c;
                    if (disc < 0.0f) continue;
                    test_candidate<DEBUG>(sc, L, polish, i, a, ocx, ocy, ocz, b, c, disc, dx, dy, dz, bt, bid, cnt);
                }
            }
        }
    }
    // [census: flat_end @ ITER]
    if (mine && !(WALK_REPLAY && replay)) {
        offer_plane<DEBUG>(oy, dy, n, bt, bid, cnt);
        best_t = bt; best_id = bid;
    }
    cnt.tick(HS_T_OWNER);
}

#ifdef RT_DEV_SWITCHES
// Development statistics: what a distance cut-off could save.  For a ray whose trace returned t_final
// (1e30: nothing hit), counts its kept leaf nodes, those whose entry distance lies in front of the hit
// (the only ones an ordered walk with a cut-off would have to open), the members that pass the
// bounding test and those among them whose t_bound lies in front of the hit; for shadow rays the same
// against the distance that is left to the light.
template <bool DEBUG>
DEV void dev_trace_stats(const float *sc, const float *hl, const SceneLayout &L,
                         float ox, float oy, float oz, float dx, float dy, float dz,
                         float t_final, bool shadow, float light_left, const Counters<DEBUG> &cnt)
{
    if constexpr (!DEBUG) {
        unsigned long long lo, hi;
        hierarchy_kept_leaves<DEBUG>(hl, L, ox, oy, oz, dx, dy, dz, lo, hi, cnt);
        const float a = dot3(dx, dy, dz, dx, dy, dz);
        const NodeTest nt = node_test_constants(L, ox, oy, oz, a);
        const float *leafn = hl + (L.off_leaf - L.off_top);
        const float4 *memb = reinterpret_cast<const float4 *>(hl + (L.off_member - L.off_top));
        const unsigned short *slot_id = reinterpret_cast<const unsigned short *>(hl + (L.off_slot_id - L.off_top));
        cnt.hadd(HS_RAYS, 1);
        if (shadow) cnt.hadd(HS_RAYS_SHADOW, 1);
        if (t_final < 1e30f) cnt.hadd(HS_RAYS_HIT, 1);
        if (t_final < 1.0f) cnt.hadd(HS_RAYS_NEAR_HIT, 1);
        const float cut = t_final + 4.0f * nt.grow;
        float first_lb = 1e30f;
        int needed = 0;
        while (lo | hi) {
            int leaf;
            if (lo) { leaf = __builtin_ctzll(lo); lo &= lo - 1; }
            else { leaf = 64 + __builtin_ctzll(hi); hi &= hi - 1; }
            const float *p = leafn + 8 * (leaf >> 1) + (leaf & 1);
            const float ocx = ox - p[0], ocy = oy - p[2], ocz = oz - p[4], Re = __builtin_fabsf(p[6]) + nt.grow;
            const float b = ocz * dz + ocy * dy + ocx * dx;
            const float sdisc = b * b - ((ocz * ocz + ocy * ocy + ocx * ocx) - Re * Re) + nt.slack;
            const float lb = -b - __builtin_amdgcn_sqrtf(sdisc > 0.0f ? sdisc : 0.0f) - nt.grow;
            cnt.hadd(HS_KEPT, 1);
            const bool need = !(lb > cut);
            if (need) { cnt.hadd(HS_KEPT_NEEDED, 1); needed++; }
            if (shadow && !(lb > cut) && !(lb > light_left + 4.0f * nt.grow)) cnt.hadd(HS_KEPT_SHADOW_NEEDED, 1);
            else if (!shadow && need) cnt.hadd(HS_KEPT_SHADOW_NEEDED, 1);
            first_lb = lb < first_lb ? lb : first_lb;
            for (int m = 0; m < RT_LEAF; ++m) {
                const int sl = leaf * RT_LEAF + m;
                const float *q = reinterpret_cast<const float *>(memb) + 8 * (sl >> 1) + (sl & 1);
                const float mx = ox - q[0], my = oy - q[2], mz = oz - q[4];
                const float c = ((mz * mz + my * my) + mx * mx) - q[6];
                const float bb = (mz * dz + my * dy) + mx * dx;
                const float disc = bb * bb - c;
                const unsigned int sid = slot_id[sl];
                if (!(disc < 0.0f) && !(!(sid & 0x8000u) && bb > 0.0f && bb < 1e15f && c >= 0.0f)) {
                    cnt.hadd(HS_CANDS, 1);
                    const float tb = -bb - __builtin_amdgcn_sqrtf(disc);
                    if (!(tb > cut)) cnt.hadd(HS_CANDS_NEEDED, 1);
                }
            }
        }
        (void)first_lb; (void)needed; (void)sc;
    }
}
#endif

// ------------------------------------------------------------------ one ray at a time, the whole wave on it
//
// For waves that have only a few rays left (the deepest pixels of a frame, after the tile queue has
// run dry): the rays are taken one after the other, and for each of them lane l runs the
// reference's bounding test (:399-402) and, where it passes, the exact intersection on objects
// l, l + 64, ... -- the same binary32 operations on the same operands as the per-lane loop, so each
// object's (t, hit) and t_bound are the same bits.  What the reference's loop does with them depends
// on the order (the t_bound > closest.t skip, "first index wins"), so the hits are then replayed in
// ascending object index on wave-uniform values.  Objects that are not hit change nothing in the
// reference's loop and are left out of the replay; so are the spheres behind the ray (see
// test_candidate).  No enclosing spheres are involved.  `mine` marks the lanes that hold a ray;
// all 64 lanes must enter.
DEV float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <bool DEBUG>
DEV void trace_cooperative(const float *sc, const SceneLayout &L, int polish, bool mine,
                           float ox, float oy, float oz, float dx, float dy, float dz,
                           float &best_t, int &best_id, const Counters<DEBUG> &cnt)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int n = L.n;
    const int *kind = reinterpret_cast<const int *>(sc + L.off_kind);
    unsigned long long rays = __ballot(mine);
    float my_t = 1e30f;
    int my_id = -1;
    while (rays) {
        const int r = __builtin_ctzll(rays);
        rays &= rays - 1;
        HSTAT(HS_COOP_RAY);
        const float rox = lane_value(ox, r), roy = lane_value(oy, r), roz = lane_value(oz, r);
        const float rdx = lane_value(dx, r), rdy = lane_value(dy, r), rdz = lane_value(dz, r);
        const float a = dot3(rdx, rdy, rdz, rdx, rdy, rdz);
        float bt = 1e30f;
        int bid = -1;
        for (int base = 0; base < n; base += 64) {
            HSTAT(HS_COOP_ROUND);
            const int i = base + lane;
            const int ic = i < n ? i : n - 1;
            const float *rec = sc + 8 * (ic >> 1) + (ic & 1);
            const float ocx = rox - rec[0], ocy = roy - rec[2], ocz = roz - rec[4];
            const float oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
            const float c = oc2 - rec[6];
            const float b = (ocz * rdz + ocy * rdy) + ocx * rdx;
            const float disc = b * b - c;
            float t = 0.0f, tb = 0.0f;
            bool hit = false;
            if (i < n && !(disc < 0.0f)) {
                const int kd = kind[ic];
                if (!(kd < 0 && b > 0.0f && b < 1e15f && c >= 0.0f)) {
                    tb = -b - rt_sqrt(disc);
                    candidate_exact<DEBUG>(sc, L, polish, kd, a, ocx, ocy, ocz, b, c, rdx, rdy, rdz, t, hit, cnt);
                }
            }
            unsigned long long hits = __ballot(hit);
            while (hits) {
                const int k = __builtin_ctzll(hits);
                hits &= hits - 1;
                const float tbk = lane_value(tb, k), tk = lane_value(t, k);
                if (bt < 1e30f && tbk > bt) continue;      // the reference skips this object (:412-415)
                if (tk < bt) { bt = tk; bid = base + k; }
            }
        }
        if (lane == r) { my_t = bt; my_id = bid; }
    }
    if (mine) {
        offer_plane<DEBUG>(oy, dy, n, my_t, my_id, cnt);
        best_t = my_t; best_id = my_id;
    }
}

}  // namespace RT_NS
