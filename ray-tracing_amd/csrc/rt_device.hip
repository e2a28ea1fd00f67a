// rt_device.hip -- the gfx950 render kernel.
//
// Compiled twice into one library:
//   strict flavour (default):  -ffp-contract=off, correctly rounded / and sqrt, the
//       reference shader's transcendental kernels restated with explicit fma.  Every FP32
//       operation is the one llvmpipe executes for fragment_shader.glsl, in its order, so the
//       frame is bit-identical to the reference's.
//   fast flavour (-DRT_FAST):  -ffp-contract=fast plus v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_cos.
// Both are the same source; the flavour only changes the primitives in the "numerics" block.
//
// Design (wave64, VALU-bound, no MFMA: there is no dense contraction in this workload):
//   * one lane = one pixel; a wave owns an 8x8 pixel tile, a 256-thread workgroup 16x16.
//   * the scene is staged once per workgroup in LDS, split so that the per-object hot loop
//     reads one 16-byte record (centre + bounding radius^2) as an LDS broadcast; torus
//     frames, kinds and materials sit behind the bounding test / the closest-hit decision.
//   * trace() keeps only (t, id) of the closest hit; position, normal and material are
//     derived once for the winner (the reference builds a full HitInfo per candidate,
//     fragment_shader.glsl:416-425; only the winner's is ever read).
//   * the reference's explicit ray stack (fragment_shader.glsl:547-564) is held in
//     registers as "current ray + pending refraction rays": the reflection child is always
//     popped right after it is pushed, so it never needs a slot; slots are addressed with
//     compare/select chains (no dynamic register indexing, no scratch memory).  Visit order
//     and the silent drop rule (stack_ptr < STACK_SIZE, :615/:632) are preserved exactly.
//   * path rays and shadow rays go through ONE trace site: each lane is a small state
//     machine (PATH / SHADOW), so lanes shading and lanes casting shadow rays share the
//     wave's trace iterations instead of serialising two inlined copies of trace().
//   * a lane that finishes a sample starts its next stratified sample immediately.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "rt_types.h"

#ifdef RT_FAST
#define RT_NS rt_fast
#else
#define RT_NS rt_strict
#endif

namespace RT_NS {

// ------------------------------------------------------------------ numerics
#define DEV __device__ __forceinline__

DEV float bits_f(uint32_t u) { return __uint_as_float(u); }
DEV uint32_t f_bits(float f) { return __float_as_uint(f); }

#ifdef RT_FAST
// fast flavour: hardware reciprocal / rsqrt / sqrt (1 ulp), FMA contraction (compile flag)
DEV float rt_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
DEV float rt_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
DEV float rt_rsq(float a) { return __builtin_amdgcn_rsqf(a); }
template <int C> DEV float rt_div_const(float x) { return x * (1.0f / (float)C); }
#else
// strict flavour: correctly rounded, as the reference's CPU GL computes them.
//
// The compiler's expansions of sqrt and of division spend a third of their instructions on
// operand scaling and on the special values (0, inf, NaN, results near the denormal range).
// For operands well inside the normal range those steps are identities, so the functions
// below run the same refinement sequences without them -- the same instructions in the same
// order, hence the same bits -- and leave every other operand to the compiler's expansion.
// rt_selftest_kernel compares both forms over every binary32 operand of the guarded ranges.
DEV float rt_div(float a, float b) { return a / b; }

// sqrt for x in [2^-96, 2^96): rsq seed, one coupled Newton step on (g ~ sqrt x, h ~ 1/(2 sqrt x)),
// one residual correction
DEV float sqrt_core(float x)
{
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r;
    float h = 0.5f * r;
    float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    float d = __builtin_fmaf(-g, g, x);
    h = __builtin_fmaf(h, e, h);
    return __builtin_fmaf(d, h, g);
}
// 1/s for s in [2^-48, 2^48]: rcp seed, Newton step, two residual corrections of the quotient
DEV float rcp_core(float s)
{
    float y = __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float r0 = __builtin_fmaf(-s, y, 1.0f);
    float q = __builtin_fmaf(r0, y, y);
    float r1 = __builtin_fmaf(-s, q, 1.0f);
    return __builtin_fmaf(r1, y, q);
}
// a / b for b in [2^-48, 2^48] and a == 0 or |a| in [2^-48, 2^48]
DEV float div_core(float a, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = a * y;
    float r0 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r0, y, q);
    float r1 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r1, y, q);
}
// x / 3 and x / 27 (the cubic's constants, :201-215) for |x| in [2^-96, 2^96): product with the
// rounded reciprocal, exact remainder, one correction.  For these two divisors the sequence
// returns the correctly rounded quotient for every such x (rt_selftest_kernel checks all of
// them against the general division).
template <int C> DEV float div_const_core(float x)
{
    constexpr float c = (float)C;
    constexpr float rc = C == 3 ? 0x1.555556p-2f : 0x1.2f684cp-5f;     // RN(1/3), RN(1/27)
    float q = x * rc;
    float r = __builtin_fmaf(-c, q, x);
    return __builtin_fmaf(r, rc, q);
}
DEV bool div_const_in_core_range(float x) { return ((f_bits(x) & 0x7fffffffu) - 0x0f800000u) < 0x60000000u; }
template <int C> DEV float rt_div_const(float x)
{
    if (div_const_in_core_range(x)) return div_const_core<C>(x);
    return x / (float)C;
}
// binary32 patterns 0x0f800000 .. 0x6f7fffff are the positive numbers in [2^-96, 2^96)
DEV bool sqrt_in_core_range(float x) { return (f_bits(x) - 0x0f800000u) < 0x60000000u; }
DEV float rt_sqrt(float a)
{
    if (sqrt_in_core_range(a)) return sqrt_core(a);
    return __builtin_sqrtf(a);
}
DEV float rt_rsq(float a)
{
    if (sqrt_in_core_range(a)) return rcp_core(sqrt_core(a));
    return 1.0f / __builtin_sqrtf(a);
}
#endif

// pow(x, y) = exp2(y * log2 x) with the polynomial kernels the reference's CPU GL runs
// (mantissa/exponent split + degree-4 minimax in ((m-1)/(m+1))^2; floor/fraction split +
// degree-5 minimax), multiply-adds fused.  Both flavours use them: the reference's pow is
// ~2 ulp off and its acos 7e-5 off, and the torus solver amplifies such differences, so a
// more accurate pow would move torus pixels away from the reference.
DEV float lp_log2(float x)
{
    uint32_t i = f_bits(x);
    float e = (float)((int)((i & 0x7f800000u) >> 23) - 127);
    float m = bits_f((i & 0x007fffffu) | 0x3f800000u);
#ifdef RT_FAST
    float y = rt_div(m - 1.0f, m + 1.0f);
#else
    float y = div_core(m - 1.0f, m + 1.0f);      // numerator 0 or in [2^-23, 1), denominator in [2, 3)
#endif
    float z = y * y;
    float z2 = z * z;
    float even = __builtin_fmaf(z2, __builtin_fmaf(z2, 0x1.a07ab2p-2f, 0x1.27a642p-1f), 0x1.715476p+1f);
    float odd = __builtin_fmaf(z2, 0x1.9d062cp-2f, 0x1.ec6ff2p-1f);
    float p = __builtin_fmaf(odd, z, even);
    float r = __builtin_fmaf(y, p, e);
    if (x >= __builtin_inff()) r = __builtin_inff();
    if (x == 0.0f) r = -__builtin_inff();
    if (!(x >= 0.0f)) r = __builtin_nanf("");
    return r;
}
DEV float rt_pow(float x, float y)
{
    if (!(x < 0.0f || x > 0.0f)) return 0.0f;
    float t = lp_log2(x) * y;              // NaN for x < 0, and NaN passes through the clamps
    t = (128.0f < t) ? 128.0f : t;
    t = (-0x1.fbfffep+6f > t) ? -0x1.fbfffep+6f : t;
    float fl = __builtin_floorf(t);
    float fpart = t - fl;
    int ipart = (int)fl;
    float scale = bits_f((uint32_t)(ipart + 127) << 23);
    float x2 = fpart * fpart;
    float odd = __builtin_fmaf(x2, __builtin_fmaf(x2, 0x1.ec320ap-10f, 0x1.c95446p-5f), 0x1.62e4f6p-1f);
    float even = __builtin_fmaf(x2, __builtin_fmaf(x2, 0x1.26900cp-7f, 0x1.ebd5a8p-3f), 1.0f);
    float p = __builtin_fmaf(odd, fpart, even);
    return scale * p;
}
// cos(x): octant reduction with a three-part pi/4 and the sin/cos minimax pair.
DEV float rt_cos(float x)
{
    float xa = __builtin_fabsf(x);
    float sy = xa * 0x1.45f306p+0f;
    int j = (int)sy;
    j = (j + 1) & ~1;
    float yf = (float)j;
    int emm2 = j - 2;
    uint32_t sign_bit = ((uint32_t)(~emm2) & 4u) << 29;
    bool use_sin = (emm2 & 2) == 0;
    float r = __builtin_fmaf(yf, -0.78515625f, xa);
    r = __builtin_fmaf(yf, -0x1.fb4p-13f, r);
    r = __builtin_fmaf(yf, -0x1.4442d2p-25f, r);
    float z = r * r;
    float s = __builtin_fmaf(z, -0x1.9943f2p-13f, 0x1.11073cp-7f);
    s = __builtin_fmaf(s, z, -0x1.555546p-3f);
    s = s * z;
    s = __builtin_fmaf(s, r, r);
    float c = __builtin_fmaf(z, 0x1.99eb9cp-16f, -0x1.6c0c34p-10f);
    c = __builtin_fmaf(c, z, 0x1.55554ap-5f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    float res = bits_f(f_bits(use_sin ? s : c) ^ sign_bit);
    if ((f_bits(x) & 0x7f800000u) == 0x7f800000u) return __builtin_nanf("");
    res = (res < 1.0f) ? res : 1.0f;
    res = (res > -1.0f) ? res : -1.0f;
    return res;
}

// acos as the GLSL front end lowers it (both flavours; it is six cheap ops and a sqrt).
DEV float rt_acos(float x)
{
    const float half_pi = 0x1.921fb6p+0f;
    float s = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    float ax = __builtin_fabsf(x);
    float t = rt_sqrt(1.0f - ax);
    float p = ax * -0x1.8334bep-6f;
    p = 0x1.4d1b0ep-4f + p;
    p = ax * p;
    p = -0x1.b78128p-3f + p;
    p = ax * p;
    p = half_pi + p;
    float q = t * p;
    q = half_pi - q;
    q = s * q;
    return half_pi - q;
}

DEV float sign_f(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
// max(0, x) with NaN -> 0, as the reference's max(0.0, x) behaves on its CPU GL
DEV float max0(float x) { return x > 0.0f ? x : 0.0f; }
// dot product in the reference's summation order: (z + y) + x
DEV float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (az * bz + ay * by) + ax * bx;
}

// ------------------------------------------------------------------ event counters
// Two uses of the same LDS counter array in the instrumented flavour: event counts (one per
// lane and event, the FLOP accounting), or -- wave_mode -- a profile of the code blocks a wave
// executes (WCNT: once per wave that enters the block with any lane; LCNT: lanes in it).
template <bool DEBUG> struct Counters {
    unsigned int *lds_ev;
    DEV void add(int) const {}
    DEV void wave(int) const {}
    DEV void lanes(int) const {}
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
};
#define CNT(ev) cnt.add(ev)
#define WCNT(slot) cnt.wave(slot)
#define LCNT(slot) cnt.lanes(slot)
enum { WP_ITER, WP_ITER_LANES, WP_PAIR, WP_CAND, WP_CAND_LANES, WP_TBOUND, WP_SPHERE, WP_SPHERE_LANES,
       WP_SPHERE_ROOT2, WP_TORUS, WP_TORUS_LANES, WP_PLANE_DIV, WP_PATH, WP_PATH_LANES, WP_SKY, WP_HIT,
       WP_HIT_LANES, WP_NORMAL_SPHERE, WP_NORMAL_TORUS, WP_FRESNEL_POW, WP_FRESNEL_DIV, WP_LOCAL, WP_SHADOW,
       WP_SHADOW_LANES, WP_SPAWN, WP_SPAWN_LANES, WP_PHONG, WP_PHONG_LANES, WP_CHILDREN, WP_REFR, WP_REFR_PUSH,
       WP_REFL, WP_POP, WP_POP_LOAD, WP_FINISH, WP_START, WP_COUNT };
static_assert(WP_COUNT <= EV_COUNT, "the wave profile shares the event counter array");

// ------------------------------------------------------------------ torus (fragment_shader.glsl:250-375)
#define Q_EPS 1e-7f

struct RootMin {
    float t; bool found;
    DEV void offer(float r) { if (r > 0.001f && r < t) { t = r; found = true; } }
};

// solve_quadratic (:159-180) for the monic calls of solve_quartic: a = 1.
// bb is b*b as the compiled reference has it.  Offers (root - offset) to `rm`.
template <bool DEBUG>
DEV void quad_monic_offer(float bb, float b, float c, float offset, RootMin &rm, const Counters<DEBUG> &cnt)
{
    float disc = bb - 4.0f * c;
    if (disc < -Q_EPS) { CNT(EV_QUAD_EARLY); return; }
    CNT(EV_QUAD_FULL);
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
    float a2 = a * a;
    float p = b - 0.375f * a2;
    float q = (c - (0.5f * a) * b) + (0.125f * a2) * a;
    float r = ((d - (0.25f * a) * c) + (0.0625f * a2) * b) - (0.01171875f * a2) * a2;
    float offset = 0.25f * a;
    RootMin rm; rm.t = 1e20f; rm.found = false;
    if (__builtin_fabsf(q) < Q_EPS) {
        // biquadratic (:262-276): quadratic in y^2
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
        CNT(EV_CUBIC);
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
            CNT(EV_CARDANO);
            disc = max0(disc);
            float sq = rt_sqrt(disc);
            float A = -half_q + sq;
            float B = -half_q - sq;
            float rootA = sign_f(A) * rt_pow(__builtin_fabsf(A), 0x1.555556p-2f);
            float rootB = sign_f(B) * rt_pow(__builtin_fabsf(B), 0x1.555556p-2f);
            u = (rootA - off_c) + rootB;
        } else {
            CNT(EV_TRIG);
            float rho = rt_sqrt(rt_div_const<27>(-p3));
            float x = rt_div(-half_q, rho);
            x = (x > -1.0f) ? x : -1.0f;       // NaN -> -1
            x = (x < 1.0f) ? x : 1.0f;
            float theta = rt_div_const<3>(rt_acos(x));
            float m = 2.0f * rt_sqrt(rt_div_const<3>(-pc));
            u = m * rt_cos(theta) - off_c;
        }
        if (!(u < 0.0f)) {
            CNT(EV_FERRARI_TAIL);
            float w = rt_sqrt(u);
            float term_A = 0.5f * (p + u);
            float term_B = rt_div(0.5f * q, w + Q_EPS);
            float bb = __builtin_fabsf(u);
            quad_monic_offer<DEBUG>(bb, w, term_A - term_B, offset, rm, cnt);
            quad_monic_offer<DEBUG>(bb, -w, term_A + term_B, offset, rm, cnt);
        }
    }
    t_out = rm.t;
    return rm.found;
}

struct TorusFrame { float ux, uy, uz, wx, wy, wz, vx, vy, vz, R, R2, r2, fourR2, eightR2; };

DEV TorusFrame load_torus(const float *lds, int off_torus, int slot)
{
    const float4 *q = reinterpret_cast<const float4 *>(lds + off_torus + 16 * slot);
    float4 a = q[0], b = q[1], c = q[2], d = q[3];
    TorusFrame f;
    f.ux = a.x; f.uy = a.y; f.uz = a.z; f.wx = a.w;
    f.wy = b.x; f.wz = b.y; f.vx = b.z; f.vy = b.w;
    f.vz = c.x; f.R = c.y; f.R2 = c.z; f.r2 = c.w;
    f.fourR2 = d.x; f.eightR2 = d.y;
    return f;
}

// Two Newton steps on the selected root; kept only if the residual shrinks.
DEV float newton_polish(float A, float B, float C, float D, float t)
{
    float t0 = t;
    float f0 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(t, 1.0f, A), t, B), t, C), t, D);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float f = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(t + A, t, B), t, C), t, D);
        float df = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(4.0f, t, 3.0f * A), t, 2.0f * B), t, C);
        float step = f * __builtin_amdgcn_rcpf(df);
        t = t - step;
    }
    float f1 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(t + A, t, B), t, C), t, D);
    bool ok = __builtin_fabsf(f1) < __builtin_fabsf(f0) && t > 0.001f;
    return ok ? t : t0;
}

// intersect_torus (:309-352) up to the root; ocx.. = ray origin - centre.
template <bool DEBUG>
DEV bool torus_nearest(const TorusFrame &f, float ocx, float ocy, float ocz, float dx, float dy, float dz,
                       int polish, float &t_out, const Counters<DEBUG> &cnt)
{
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
    if (found && polish) t = newton_polish(A, B, C, D, t);
    t_out = t;
    return found;
}

// ------------------------------------------------------------------ trace (fragment_shader.glsl:379-435)
typedef float v2f __attribute__((ext_vector_type(2)));

// Centre of object `id` from the pair-interleaved bound records.
DEV void object_centre(const float *lds, int id, float &cx, float &cy, float &cz)
{
    const float *p = lds + 8 * (id >> 1) + (id & 1);
    cx = p[0]; cy = p[2]; cz = p[4];
}

// Exact test of one object whose bounding test did not reject it (:405-425).
template <bool DEBUG>
DEV void test_candidate(const float *lds, const SceneLayout &L, int polish, int i, float a,
                        float ocx, float ocy, float ocz, float b, float c, float disc,
                        float dx, float dy, float dz, float &bt, int &bid, const Counters<DEBUG> &cnt)
{
    CNT(EV_BOUND_TBOUND);
    WCNT(WP_CAND); LCNT(WP_CAND_LANES);
    // t_bound > closest.t can only hold once something has been hit: while closest.t is
    // still 1e30 the comparison is false for every finite ray, so the sqrt is skipped.
    if (bt < 1e30f) {
        WCNT(WP_TBOUND);
        float t_bound = -b - rt_sqrt(disc);
        if (t_bound > bt) return;
    }
    int kd = reinterpret_cast<const int *>(lds + L.off_kind)[i];
    float t; bool hit;
    if (kd < 0) {
        // intersect_sphere (:74-92); c is the same dot(oc,oc) - r*r
        CNT(EV_SPHERE_CALL);
        WCNT(WP_SPHERE); LCNT(WP_SPHERE_LANES);
        float b2 = 2.0f * b;
        float ds = b2 * b2 - (4.0f * a) * c;
        hit = false; t = 0.0f;
        if (!(ds < 0.0f)) {
            float sq = rt_sqrt(ds);
            float two_a = 2.0f * a;
            CNT(EV_SPHERE_ROOT);
            t = rt_div(-b2 - sq, two_a);
            if (t < 0.0f) { CNT(EV_SPHERE_ROOT); WCNT(WP_SPHERE_ROOT2); t = rt_div(-b2 + sq, two_a); }
            hit = t > 0.001f;
            if (hit) CNT(EV_SPHERE_HIT);
        }
    } else {
        WCNT(WP_TORUS); LCNT(WP_TORUS_LANES);
        TorusFrame f = load_torus(lds, L.off_torus, kd);
        hit = torus_nearest<DEBUG>(f, ocx, ocy, ocz, dx, dy, dz, polish, t, cnt);
        if (hit) CNT(EV_TORUS_HIT);
    }
    if (hit && t < bt) { bt = t; bid = i; }
}

// intersect_plane (:107-123): y = -1, two-sided; tested after all objects (:428-433).
template <bool DEBUG>
DEV void offer_plane(float oy, float dy, int n, float &bt, int &bid, const Counters<DEBUG> &cnt)
{
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
    CNT(EV_TRACE);
    float bt = 1e30f;
    int bid = -1;
    const float4 *pairs = reinterpret_cast<const float4 *>(lds);
    const float a = dot3(dx, dy, dz, dx, dy, dz);    // intersect_sphere's dot(d,d) (:78)
    const int n = L.n;
    const int npairs = (n + 1) >> 1;
    const v2f OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
    const v2f DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
    float4 A = make_float4(0, 0, 0, 0), B = A;
    if (npairs > 0) { A = pairs[0]; B = pairs[1]; }
    for (int j = 0; j < npairs; ++j) {
        WCNT(WP_PAIR);
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
    offer_plane<DEBUG>(oy, dy, n, bt, bid, cnt);
    best_t = bt; best_id = bid;
}

// ------------------------------------------------------------------ trace through the sphere hierarchy
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
// The candidates are queued per ray in LDS as 16-bit object indices; a ray that collects
// more than RT_QUEUE of them walks all objects in step 2 instead (same result, slower).
DEV void node_keep(float4 A, float4 B, v2f OX, v2f OY, v2f OZ, v2f DX, v2f DY, v2f DZ, float grow, float slack,
                   bool &keep0, bool &keep1)
{
    v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, R = {B.z, B.w};
    v2f ocx = OX - cx, ocy = OY - cy, ocz = OZ - cz;
    v2f oc2 = (ocz * ocz + ocy * ocy) + ocx * ocx;
    v2f G = {grow, grow}, S = {slack, slack};
    v2f Re = R + G;
    v2f b = (ocz * DZ + ocy * DY) + ocx * DX;
    v2f disc = b * b - (oc2 - Re * Re);
    v2f s = disc + S;
    keep0 = !(s.x < 0.0f);
    keep1 = !(s.y < 0.0f);
}

template <bool DEBUG>
DEV void trace_hierarchy(const float *sc, const float *hl, const SceneLayout &L, int polish, unsigned short *queue,
                         float ox, float oy, float oz, float dx, float dy, float dz,
                         float &best_t, int &best_id, const Counters<DEBUG> &cnt)
{
    const float a = dot3(dx, dy, dz, dx, dy, dz);    // intersect_sphere's dot(d,d) (:78)
    const int n = L.n, n1 = L.n1, n2 = L.n2;
    const v2f OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
    const v2f DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
    // growth and slack of the node test, once per ray: every object and node lies inside the
    // scene sphere (L.hc, L.hR), so |oc| + R <= reach = |o - hc| + hR for all of them
    float sf2 = 2.5e-6f + 1.1f * __builtin_fabsf(a - 1.0f);
    const float sf = __builtin_amdgcn_sqrtf(sf2) * 1.02f;
    const float qx = ox - L.hc[0], qy = oy - L.hc[1], qz = oz - L.hc[2];
    const float reach = (__builtin_amdgcn_sqrtf((qz * qz + qy * qy) + qx * qx) + L.hR) * 1.0001f;
    const float grow = sf * reach;
    const float slack = (sf2 * 1.05f) * (reach * reach + (reach + grow) * (reach + grow)) + 1e-30f;
    // hl = the hierarchy sections in LDS, starting at the top nodes
    const float4 *top = reinterpret_cast<const float4 *>(hl);
    const float4 *leafn = reinterpret_cast<const float4 *>(hl + (L.off_leaf - L.off_top));
    const float4 *memb = reinterpret_cast<const float4 *>(hl + (L.off_member - L.off_top));
    const unsigned short *slot_id = reinterpret_cast<const unsigned short *>(hl + (L.off_slot_id - L.off_top));

    // ---- step 1: collect the objects whose bounding test passes.
    // The top nodes are tested by all lanes together; below them every lane walks its own
    // list of kept nodes (bit masks), so a wave of incoherent rays pays for the longest list
    // of any lane, not for the union of all lists.
    int count = 0;
    unsigned int top_mask = 0;
    for (int j = 0; j < ((n2 + 1) >> 1); ++j) {
        bool k0, k1;
        node_keep(top[2 * j], top[2 * j + 1], OX, OY, OZ, DX, DY, DZ, grow, slack, k0, k1);
        top_mask |= (k0 ? 1u : 0u) << (2 * j);
        top_mask |= ((k1 && 2 * j + 1 < n2) ? 2u : 0u) << (2 * j);
    }
    unsigned long long leaf_lo = 0, leaf_hi = 0;        // kept leaf nodes 0..63 / 64..127
    const int last_leaf_pair = (n1 - 1) >> 1;
    while (top_mask) {
        const int tn = __builtin_ctz(top_mask);
        top_mask &= top_mask - 1;
        const int first = tn * RT_FAN;                       // first leaf node of this top node
        const int nleaf = (n1 - first < RT_FAN) ? n1 - first : RT_FAN;
        unsigned int bits = 0;
#pragma unroll
        for (int kk = 0; kk < RT_FAN / 2; ++kk) {
            int lp = (first >> 1) + kk;
            lp = lp < last_leaf_pair ? lp : last_leaf_pair;   // stay inside the section; masked below
            bool k0, k1;
            node_keep(leafn[2 * lp], leafn[2 * lp + 1], OX, OY, OZ, DX, DY, DZ, grow, slack, k0, k1);
            bits |= ((k0 && 2 * kk < nleaf) ? 1u : 0u) << (2 * kk);
            bits |= ((k1 && 2 * kk + 1 < nleaf) ? 2u : 0u) << (2 * kk);
        }
        const unsigned long long placed = (unsigned long long)bits << (first & 63);
        if (first < 64) leaf_lo |= placed; else leaf_hi |= placed;
    }
    while (leaf_lo | leaf_hi) {
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
            if (!(disc.x < 0.0f)) {
                if (count < RT_QUEUE) queue[count * 256] = slot_id[2 * mp];
                count++;
            }
            if (!(disc.y < 0.0f)) {
                if (count < RT_QUEUE) queue[count * 256] = slot_id[2 * mp + 1];
                count++;
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
                int v = queue[q * 256];
                mnext = (v > last && v < mnext) ? v : mnext;
            }
            if (mnext == 0x7fffffff) break;
            i = mnext;
        } else {
            i = last + 1;
            if (i >= n) break;
        }
        last = i;
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

// ------------------------------------------------------------------ pending-ray slots
template <int CAP> struct Pending {
    float ox[CAP], oy[CAP], oz[CAP], dx[CAP], dy[CAP], dz[CAP], tx[CAP], ty[CAP], tz[CAP], ior[CAP];
    int depth[CAP];
};

#define SEL_STORE(arr, val) arr[k] = (k == slot) ? (val) : arr[k]
#define SEL_LOAD(dst, arr) dst = (k == slot) ? arr[k] : dst

// ------------------------------------------------------------------ the kernel
enum { ST_PATH = 0, ST_SHADOW = 1 };

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 2
#endif
#ifndef RT_WAVES_PER_EU_HIER
#define RT_WAVES_PER_EU_HIER 3
#endif
#ifndef RT_WAVES_PER_EU_LDSP
#define RT_WAVES_PER_EU_LDSP 4
#endif
// Template flags of the kernel:
// MULTI: more than one sample per pixel (sample loop + accumulators); the single-sample
// kernel carries neither.
// HIER: trace through the sphere hierarchy (large scenes); only the hierarchy sections and the
// per-lane candidate queues are in LDS, the rest of the scene is read from global memory.
// REFILL: persistent waves behind a work queue.  The frame is a queue of 8x8 tiles in
// RT_SUBQUEUES interleaved sub-queues; a wave takes a tile, and whenever lanes finish their
// pixel they are handed the next unassigned pixels of the wave's current tile (ballot + prefix
// count), a new tile being taken when it runs out.  Path-tree sizes differ by orders of
// magnitude between neighbouring pixels at depth 16, so without this most lanes of a wave
// wait for its longest pixel.  With kp.whole_tiles the pixels are handed out only when the
// whole wave is idle: tile after tile with the lanes in lock-step (small scenes).
// PEND: where the pending-ray slots live: PEND_REG registers (CAP of them, compare/select
// addressing), PEND_LDS ([slot][field][thread] after the scene), PEND_GLOBAL (same layout in a
// per-workgroup slice of a global workspace; persistent-wave kernels only, whose grid is
// bounded by the number of resident workgroups), PEND_HYBRID (the first three slots in LDS,
// deeper ones in the workspace: most paths never go deeper, and the LDS footprint stays that
// of the three-slot kernel).
enum { PEND_REG = 0, PEND_LDS = 1, PEND_GLOBAL = 2, PEND_HYBRID = 3 };
#define RT_HYBRID_LDS_SLOTS 3      // PEND_HYBRID: slots 0..2 in LDS, the rest in the global workspace
template <int CAP, bool DEBUG, int PEND, bool MULTI, bool HIER, bool REFILL>
__global__ __launch_bounds__(256, (PEND == PEND_LDS || PEND == PEND_HYBRID) ? RT_WAVES_PER_EU_LDSP : (HIER && (CAP <= 7 || PEND == PEND_GLOBAL)) ? RT_WAVES_PER_EU_HIER : RT_WAVES_PER_EU)
void rt_kernel(KParams kp, const float *__restrict__ scene, void *__restrict__ out,
               unsigned long long *__restrict__ ev_out, float *__restrict__ cap_out)
{
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    const SceneLayout L = kp.scene;
    // HIER keeps only the hierarchy sections in LDS (they are what every ray reads many
    // times); bound records, materials and torus frames are read a few times per ray and
    // come from global memory through the caches, which leaves LDS room for several
    // workgroups per CU however large the scene is.
    const int lds_first = HIER ? L.off_top : 0;
    const int lds_dwords = L.total_dwords - lds_first;
    {
        const float4 *src = reinterpret_cast<const float4 *>(scene + lds_first);
        const int nq = lds_dwords >> 2;
        for (int i = threadIdx.x; i < nq; i += 256) lds4[i] = src[i];
    }
    const float *sc = HIER ? scene : lds;       // bound records, materials, torus frames, kinds
    Counters<DEBUG> cnt;
    cnt.lds_ev = reinterpret_cast<unsigned int *>(lds + lds_dwords);
    if constexpr (DEBUG) cnt.wave_mode = kp.wave_profile;
    if (DEBUG) { if (threadIdx.x < EV_COUNT) cnt.lds_ev[threadIdx.x] = 0u; }
    __syncthreads();

    // ---- which pixel
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int px, packed_row, gy;     // pixel column, row inside this part's buffer, GL row (0 = bottom)
    bool active;
    if (!REFILL) {
        px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
        const int band = kp.part + blockIdx.y * kp.nparts;
        const int in_band = (wave >> 1) * 8 + (lane >> 3);
        const int row = band * 16 + in_band;             // output row
        packed_row = blockIdx.y * 16 + in_band;
        active = px < kp.w && row < kp.h;
        gy = kp.bottom_up ? row : (kp.h - 1 - row);
    } else {
        px = 0; packed_row = 0; gy = 0; active = false;
    }

    // SYNC: with several samples per pixel and one pixel per lane, a lane that has finished a
    // sample waits for the rest of its wave before it starts the next one: lanes that run ahead
    // leave lock-step and then every code block of the loop executes on every iteration
    constexpr bool SYNC = MULTI && !REFILL && !DEBUG;
    bool parked = false;
    const int N = MULTI ? kp.spp : 1;
    const int nsamples = N * N;
    const float inv_spp = 1.0f / (float)nsamples;

    // ---- per-lane state
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 1, dz = 0;  // ray being traced
    float thx = 1, thy = 1, thz = 1, cur_ior = 1;          // path segment: throughput, medium
    int depth = 0;
    float fr = 0, fg = 0, fb = 0;                          // colour of the current sample
    float accr = 0, accg = 0, accb = 0;                    // sum over samples
    int sample = 0;
    int np = 0;                                            // pending refraction rays
    constexpr bool LDSP = PEND == PEND_LDS || PEND == PEND_HYBRID;
    constexpr bool MEMP = PEND != PEND_REG;             // slots in memory, not registers
    Pending<MEMP ? 1 : CAP> pend;
#pragma unroll
    for (int k = 0; k < (MEMP ? 1 : CAP); ++k) {
        pend.ox[k] = pend.oy[k] = pend.oz[k] = 0; pend.dx[k] = pend.dy[k] = pend.dz[k] = 0;
        pend.tx[k] = pend.ty[k] = pend.tz[k] = 0; pend.ior[k] = 1; pend.depth[k] = 0;
    }
    // LDS pending slots start after the scene (and the debug counters), 16-byte aligned
    float *pend_lds = lds + ((lds_dwords + (DEBUG ? EV_COUNT : 0) + 3) & ~3) + threadIdx.x;
    float *pend_glb = kp.pend_ws + (size_t)blockIdx.x * (RT_PEND_MAX * 11 * 256) + threadIdx.x;
    float *pend_mem = PEND == PEND_GLOBAL ? pend_glb : pend_lds;
    (void)LDSP;
    // candidate queue of the hierarchy walk: [entry][thread], 16 bits each (never together with LDSP)
    unsigned short *queue = reinterpret_cast<unsigned short *>(lds + ((lds_dwords + (DEBUG ? EV_COUNT : 0) + 3) & ~3)) + threadIdx.x;
    int state = ST_PATH;
    // shading context carried across the shadow hops of one hit
    float hx = 0, hy = 0, hz = 0;            // hit position
    float nx = 0, ny = 1, nz = 0;            // geometric normal
    float pdx = 0, pdy = 0, pdz = 0;         // direction of the path ray that hit
    float mr = 0, mg = 0, mb = 0, m_refl = 0, m_transp = 0;
    float osign = 1, dni = 0, eta = 1, n2 = 1, F = 0, local = 0;
    float light_dist = 0, travelled = 0, filr = 1, filg = 1, filb = 1;
    float stx = 0, sty = 0, stz = 0;         // shadow step = L * 0.001
    int hop = 0;
    // debug capture
    int seg = 0, cap_nseg = 0; float cap_id = -2.0f, cap_t = 0.0f, cap_depth = 0.0f;

    // stratified sample positions: offset of sample column/row i from the pixel centre,
    // (i + 0.5)/N - 0.5 rounded once to binary32 (host-computed table, kp.sample_off)
    int si = 0, sj = 0;
    auto start_sample = [&]() {
        // main() (:650-668): pixel centre + stratified offset
        float fx = (float)px + 0.5f, fy = (float)gy + 0.5f;
        if (MULTI) { fx = fx + kp.sample_off[si]; fy = fy + kp.sample_off[sj]; }
        float uvx = rt_div(fx * 2.0f - kp.res_w, kp.res_h);
        float uvy = rt_div(fy * 2.0f - kp.res_h, kp.res_h);
        float rx = (uvx * kp.right[0] + uvy * kp.up[0]) + kp.focal * kp.fwd[0];
        float ry = (uvx * kp.right[1] + uvy * kp.up[1]) + kp.focal * kp.fwd[1];
        float rz = (uvx * kp.right[2] + uvy * kp.up[2]) + kp.focal * kp.fwd[2];
        float rs = rt_rsq(dot3(rx, ry, rz, rx, ry, rz));
        dx = rx * rs; dy = ry * rs; dz = rz * rs;
        ox = kp.eye[0]; oy = kp.eye[1]; oz = kp.eye[2];
        thx = thy = thz = 1.0f; cur_ior = 1.0f; depth = 0;
        fr = fg = fb = 0.0f; np = 0; state = ST_PATH; seg = 0;
        CNT(EV_PIXEL_PATH);
        WCNT(WP_START);
    };
    // RGBA8 UNORM conversion of the framebuffer write (:674): clamp, *255, round to even
    auto write_pixel = [&]() {
        float cr = fr, cg = fg, cb = fb;
        if (MULTI) { cr = accr * inv_spp; cg = accg * inv_spp; cb = accb * inv_spp; }
        size_t o = (size_t)packed_row * kp.w + px;
        if (kp.out_f32) {
            reinterpret_cast<float4 *>(out)[o] = make_float4(cr, cg, cb, 1.0f);
        } else {
            auto u8 = [](float c) -> uint32_t {
                float s = !(c > 0.0f) ? 0.0f : (c >= 1.0f ? 1.0f : c);
                return (uint32_t)__builtin_rintf(s * 255.0f);
            };
            uint32_t v = u8(cr) | (u8(cg) << 8) | (u8(cb) << 16) | 0xff000000u;
            reinterpret_cast<uint32_t *>(out)[o] = v;
        }
    };

    const bool in_frame = active;       // fixed pixel per lane (without REFILL)
    // wave-uniform queue state (REFILL): the tile being handed out and its next unassigned pixel
    const int tiles_x8 = (kp.w + 7) >> 3;
    const int ntiles = tiles_x8 * (kp.my_bands * 2);
    int pool_tile = -1, pool_next = 64;
    int sq = (int)((blockIdx.x * 4u + (unsigned int)wave) % RT_SUBQUEUES), sq_tried = 0;
    bool exhausted = false;
    if (!REFILL && active) start_sample();

    for (;;) {
        if (!REFILL) {
            if (SYNC) {
                // lanes start sample k+1 together, once the whole wave has finished sample k
                if (__ballot(active && !parked) == 0ull) {
                    if (__ballot(active) == 0ull) break;
                    if (active) { start_sample(); parked = false; }
                }
            } else if (!active) break;
        } else {
            const unsigned long long need = __ballot(!active);
            // whole_tiles: pixels are handed out only when the whole wave is idle, i.e. a wave
            // renders tile after tile with its lanes in lock-step (small scenes)
            if (need != 0ull && (!kp.whole_tiles || need == ~0ull) && (pool_next < 64 || !exhausted)) {
                const int nneed = __popcll(need);
                const int my = __builtin_amdgcn_mbcnt_hi((unsigned int)(need >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned int)need, 0u));
                const bool cross = pool_next + nneed > 64;
                int new_tile = -1;
                if (cross && !exhausted) {
                    // The queue is RT_SUBQUEUES interleaved sub-queues (sub-queue q holds the tiles
                    // t = q mod RT_SUBQUEUES, in raster order), each behind its own counter: one
                    // counter for the whole GPU sustains only ~1e8 operations per second, which
                    // bounded frames of cheap tiles; interleaving keeps all waves in the same
                    // region of the frame.  A wave drains the sub-queue it starts on, then moves
                    // on to the next one that still has tiles.
                    bool got = false;
                    while (sq_tried < RT_SUBQUEUES) {
                        const int cnt = (ntiles - sq + RT_SUBQUEUES - 1) / RT_SUBQUEUES;   // tiles of sub-queue sq
                        if (cnt > 0) {
                            unsigned int tk = 0;
                            if (lane == 0) tk = atomicAdd(kp.work_counter + sq * RT_SUBQUEUE_STRIDE, 1u);
                            tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
                            if (tk < (unsigned int)cnt) { new_tile = (int)tk * RT_SUBQUEUES + sq; got = true; break; }
                        }
                        sq = sq + 1 == RT_SUBQUEUES ? 0 : sq + 1;
                        sq_tried++;
                    }
                    if (!got) exhausted = true;
                }
                if (!active) {
                    int idx = pool_next + my;
                    const int tile = idx < 64 ? pool_tile : new_tile;
                    idx &= 63;
                    if (tile >= 0) {
                        const int ty = tile / tiles_x8, tx = tile - ty * tiles_x8;
                        px = tx * 8 + (idx & 7);
                        packed_row = ty * 8 + (idx >> 3);
                        const int row = (kp.part + (packed_row >> 4) * kp.nparts) * 16 + (packed_row & 15);
                        if (px < kp.w && row < kp.h) {
                            gy = kp.bottom_up ? row : (kp.h - 1 - row);
                            if (MULTI) { accr = accg = accb = 0.0f; sample = 0; si = 0; sj = 0; }
                            start_sample();
                            active = true;
                        }
                    }
                }
                if (cross) { pool_tile = new_tile; pool_next = new_tile >= 0 ? pool_next + nneed - 64 : 64; }
                else pool_next += nneed;
            }
            if (__ballot(active) == 0ull) {
                if (exhausted && pool_next >= 64) break;
                continue;
            }
        }
        // One trace + shading step of the lanes that have a pixel.  Idle lanes skip forward to
        // the end of the loop body (never `continue`: a back edge taken by idle lanes alone
        // would let them spin in a loop of their own while the busy lanes wait to reconverge).
        do {
        if (REFILL && !active) break;
        if (SYNC && (!active || parked)) break;
        WCNT(WP_ITER); LCNT(WP_ITER_LANES);
        float t; int id;
        if (HIER) trace_hierarchy<DEBUG>(sc, lds, L, kp.polish, queue, ox, oy, oz, dx, dy, dz, t, id, cnt);
        else trace_closest<DEBUG>(lds, L, kp.polish, ox, oy, oz, dx, dy, dz, t, id, cnt);

        bool spawn = false;     // this lane finished shading a hit: spawn children
        bool pop = false;       // this lane needs its next ray from the pending slots

        if (state == ST_PATH) {
            WCNT(WP_PATH); LCNT(WP_PATH_LANES);
            if (DEBUG) {
                if (sample == 0 && seg == kp.capture_seg) { cap_id = (float)id; cap_t = id >= 0 ? t : 0.0f; cap_depth = (float)depth; }
                seg++;
            }
            if (id < 0) {
                // sky (:570-580); mix(h, z, t) = h + t*(z - h); the blue channel is exactly 1
                CNT(EV_SEG_MISS);
                WCNT(WP_SKY);
                float tt = 0.5f + 0.5f * dy;
                float sx = 0.8f + tt * (0.5f - 0.8f);
                float sy = 0.9f + tt * (0.7f - 0.9f);
                fr = fr + sx * thx;
                fg = fg + sy * thy;
                fb = fb + thz;
                pop = true;
            } else {
                CNT(EV_SEG_HIT);
                WCNT(WP_HIT); LCNT(WP_HIT_LANES);
                // ---- HitInfo of the winner (:93-100, :122-141, :357-372)
                hx = ox + t * dx; hy = oy + t * dy; hz = oz + t * dz;
                float h_ior, h_r0, h_eta_in;
                if (id == L.n) {
                    nx = 0.0f; ny = 1.0f; nz = 0.0f;
                    float s = __builtin_floorf(hx) + __builtin_floorf(hz);
                    float two_fl = 2.0f * __builtin_floorf(s * 0.5f);
                    float col = (s == two_fl) ? 0.9f : 0.1f;
                    mr = mg = mb = col; m_refl = 0.05f; m_transp = 0.0f; h_ior = 1.0f;
                    h_r0 = 0.0f; h_eta_in = 1.0f;         // ((1-1)/(1+1))^2 and 1/1
                } else {
                    const float4 *mat = reinterpret_cast<const float4 *>(sc + L.off_material) + 2 * id;
                    float4 m0 = mat[0], m1 = mat[1];
                    mr = m0.x; mg = m0.y; mb = m0.z; m_refl = m0.w; m_transp = m1.x; h_ior = m1.y;
                    h_r0 = m1.z; h_eta_in = m1.w;
                    float qx, qy, qz;
                    object_centre(sc, id, qx, qy, qz);
                    int kd = reinterpret_cast<const int *>(sc + L.off_kind)[id];
                    if (kd < 0) {
                        WCNT(WP_NORMAL_SPHERE);
                        float vx = hx - qx, vy = hy - qy, vz = hz - qz;
                        float rs = rt_rsq(dot3(vx, vy, vz, vx, vy, vz));
                        nx = vx * rs; ny = vy * rs; nz = vz * rs;
                    } else {
                        WCNT(WP_NORMAL_TORUS);
                        TorusFrame f = load_torus(sc, L.off_torus, kd);
                        float ocx = ox - qx, ocy = oy - qy, ocz = oz - qz;
                        float rox = dot3(ocx, ocy, ocz, f.ux, f.uy, f.uz);
                        float roy = dot3(ocx, ocy, ocz, f.wx, f.wy, f.wz);
                        float roz = dot3(ocx, ocy, ocz, f.vx, f.vy, f.vz);
                        float rdx = dot3(dx, dy, dz, f.ux, f.uy, f.uz);
                        float rdy = dot3(dx, dy, dz, f.wx, f.wy, f.wz);
                        float rdz = dot3(dx, dy, dz, f.vx, f.vy, f.vz);
                        float lx = rox + t * rdx, ly = roy + t * rdy, lz = roz + t * rdz;
                        float alpha = rt_div(f.R, rt_sqrt(lx * lx + lz * lz));
                        float g = 1.0f - alpha;
                        float ax = lx * g, ay = ly, az = lz * g;
                        float rs = rt_rsq(dot3(ax, ay, az, ax, ay, az));
                        ax = ax * rs; ay = ay * rs; az = az * rs;
                        float wx = (ax * f.ux + ay * f.wx) + az * f.vx;
                        float wy = (ax * f.uy + ay * f.wy) + az * f.vy;
                        float wz = (ax * f.uz + ay * f.wz) + az * f.vz;
                        float rs2 = rt_rsq(dot3(wx, wy, wz, wx, wy, wz));
                        nx = wx * rs2; ny = wy * rs2; nz = wz * rs2;
                    }
                }
                // ---- orientation + Fresnel (:583-597, :524-534); dot(d,n) < 0 in compiled form
                bool entering = (dz * nz + dy * ny) < -(dx * nx);
                osign = entering ? 1.0f : -1.0f;
                float n1 = entering ? cur_ior : h_ior;
                n2 = entering ? h_ior : 1.0f;
                float onx = entering ? nx : -nx, ony = entering ? ny : -ny, onz = entering ? nz : -nz;
                dni = dot3(dx, dy, dz, onx, ony, onz);
                float cos_t = __builtin_fabsf(dni);
                // r0 = ((n1-n2)/(n1+n2))^2 and eta = n1/n2 (:525-528).  Between air (1.0) and the
                // object's medium both are per-object constants, staged with the material by the
                // host with the same binary32 operations: entering from air n1 = 1, n2 = ior;
                // leaving n1 = ior, n2 = 1 (r0 is the same value both ways: the quotient only
                // changes sign, and eta = ior / 1 = ior).  Only a ray that enters from a medium
                // other than air computes them here.
                float r0 = h_r0;
                eta = entering ? h_eta_in : h_ior;
                if (entering && cur_ior != 1.0f) {
                    WCNT(WP_FRESNEL_DIV);
                    r0 = rt_div(n1 - n2, n1 + n2); r0 = r0 * r0;
                    eta = rt_div(n1, n2);
                }
                float sin_t2 = (eta * eta) * (1.0f - cos_t * cos_t);
                if (sin_t2 > 1.0f) { CNT(EV_FRESNEL_TIR); F = 1.0f; }
                else { CNT(EV_FRESNEL_FULL); WCNT(WP_FRESNEL_POW); F = r0 + (1.0f - r0) * rt_pow(1.0f - cos_t, 5.0f); }
                local = (1.0f - m_refl) - m_transp;
                pdx = dx; pdy = dy; pdz = dz;
                if (local > 0.0f) {
                    // phong_lighting -> calculate_light_attenuation set-up (:443-450)
                    CNT(EV_LOCAL); CNT(EV_ATTEN_FIXED); WCNT(WP_LOCAL);
                    float lx = kp.light[0] - hx, ly = kp.light[1] - hy, lz = kp.light[2] - hz;
                    float d2 = dot3(lx, ly, lz, lx, ly, lz);
                    light_dist = rt_sqrt(d2);
                    float rs = rt_rsq(d2);
                    dx = lx * rs; dy = ly * rs; dz = lz * rs;      // shadow direction L
                    stx = dx * 0.001f; sty = dy * 0.001f; stz = dz * 0.001f;
                    ox = hx + stx; oy = hy + sty; oz = hz + stz;
                    filr = filg = filb = 1.0f; travelled = 0.0f; hop = 0;
                    if (kp.hops > 0) state = ST_SHADOW;             // trace the first hop next
                    else spawn = true;                               // zero hops: unattenuated
                } else {
                    spawn = true;
                }
            }
        } else {
            // ---- one shadow hop came back (:453-480)
            WCNT(WP_SHADOW); LCNT(WP_SHADOW_LANES);
            bool lit_done = true;       // attenuation finished -> shade now
            if (id >= 0) {
                CNT(EV_HOP_HIT_TEST);
                float reach = t + travelled;
                if (reach < light_dist) {
                    CNT(EV_HOP_FILTER);
                    float cr, cg, cb, ctr;
                    if (id == L.n) {
                        // the ground is opaque (:128): whichever checker colour, colour * 0 = +0
                        cr = cg = cb = 0.9f; ctr = 0.0f;
                    } else {
                        const float4 *mat = reinterpret_cast<const float4 *>(sc + L.off_material) + 2 * id;
                        float4 m0 = mat[0];
                        cr = m0.x; cg = m0.y; cb = m0.z; ctr = mat[1].x;
                    }
                    filr = filr * (cr * ctr); filg = filg * (cg * ctr); filb = filb * (cb * ctr);
                    if ((filb * filb + filg * filg) == -(filr * filr)) {
                        filr = filg = filb = 0.0f;               // fully blocked (:468-470)
                    } else {
                        CNT(EV_HOP_CONTINUE);
                        travelled = reach;
                        hop++;
                        if (hop < kp.hops) {
                            ox = (ox + t * dx) + stx; oy = (oy + t * dy) + sty; oz = (oz + t * dz) + stz;
                            lit_done = false;                    // keep tracing towards the light
                        }
                    }
                }
            }
            if (lit_done) { spawn = true; state = ST_PATH; }
        }

        if (spawn) {
            WCNT(WP_SPAWN); LCNT(WP_SPAWN_LANES);
            if (local > 0.0f) {
                // phong_lighting (:488-514) with attenuation = fil*, light direction = d (still L)
                float ambr = 0.05f * mr, ambg = 0.05f * mg, ambb = 0.05f * mb;
                float lcr = ambr, lcg = ambg, lcb = ambb;
                if ((filb * filb + filg * filg) != -(filr * filr)) {
                    CNT(EV_PHONG_FULL);
                    WCNT(WP_PHONG); LCNT(WP_PHONG_LANES);
                    float Lx = dx, Ly = dy, Lz = dz;
                    float nlz = nz * Lz, nly = ny * Ly, nlx = nx * Lx;
                    float diff = max0((nlz + nly) + nlx);
                    float vx = kp.eye[0] - hx, vy = kp.eye[1] - hy, vz = kp.eye[2] - hz;
                    float rs = rt_rsq(dot3(vx, vy, vz, vx, vy, vz));
                    vx = vx * rs; vy = vy * rs; vz = vz * rs;
                    float two_dn = (((-nlz) + (-nly)) + (-nlx)) * 2.0f;
                    float rx = -Lx - two_dn * nx, ry = -Ly - two_dn * ny, rz = -Lz - two_dn * nz;
                    float sd = max0(dot3(vx, vy, vz, rx, ry, rz));
                    float spec = 0.5f * rt_pow(sd, 32.0f);
                    lcr = ambr + (diff * mr + spec) * filr;
                    lcg = ambg + (diff * mg + spec) * filg;
                    lcb = ambb + (diff * mb + spec) * filb;
                } else {
                    CNT(EV_PHONG_SHADOWED);
                }
                fr = fr + (lcr * local) * thx;
                fg = fg + (lcg * local) * thy;
                fb = fb + (lcb * local) * thz;
            }
            // ---- children (:606-643)
            pop = true;
            if (depth < kp.max_depth - 1) {
                WCNT(WP_CHILDREN);
                float onx = osign * nx, ony = osign * ny, onz = osign * nz;
                if (m_transp > 0.0f) {
                    CNT(EV_REFR_TRY);
                    WCNT(WP_REFR);
                    float k2 = 1.0f - eta * (eta * (1.0f - dni * dni));
                    float rx = 0.0f, ry = 0.0f, rz = 0.0f;
                    if (!(k2 < 0.0f)) {
                        float g = eta * dni + rt_sqrt(k2);
                        rx = eta * pdx - g * onx; ry = eta * pdy - g * ony; rz = eta * pdz - g * onz;
                    }
                    bool valid = -(rz * rz + ry * ry) < rx * rx;
                    if (valid && np < kp.stack_size) {
                        CNT(EV_REFR_PUSH);
                        WCNT(WP_REFR_PUSH);
                        float omf = 1.0f - F;
                        float qx = hx + rx * 0.001f, qy = hy + ry * 0.001f, qz = hz + rz * 0.001f;
                        float ux = ((thx * omf) * m_transp) * mr;
                        float uy = ((thy * omf) * m_transp) * mg;
                        float uz = ((thz * omf) * m_transp) * mb;
                        const int slot = np;
                        if (PEND == PEND_HYBRID && slot >= RT_HYBRID_LDS_SLOTS) {
                            float *q = pend_glb + slot * (11 * 256);
                            q[0 * 256] = qx; q[1 * 256] = qy; q[2 * 256] = qz;
                            q[3 * 256] = rx; q[4 * 256] = ry; q[5 * 256] = rz;
                            q[6 * 256] = ux; q[7 * 256] = uy; q[8 * 256] = uz;
                            q[9 * 256] = n2; q[10 * 256] = __int_as_float(depth + 1);
                        } else if (MEMP) {
                            float *q = pend_mem + slot * (11 * 256);
                            q[0 * 256] = qx; q[1 * 256] = qy; q[2 * 256] = qz;
                            q[3 * 256] = rx; q[4 * 256] = ry; q[5 * 256] = rz;
                            q[6 * 256] = ux; q[7 * 256] = uy; q[8 * 256] = uz;
                            q[9 * 256] = n2; q[10 * 256] = __int_as_float(depth + 1);
                        } else {
#pragma unroll
                            for (int k = 0; k < (MEMP ? 1 : CAP); ++k) {
                                SEL_STORE(pend.ox, qx); SEL_STORE(pend.oy, qy); SEL_STORE(pend.oz, qz);
                                SEL_STORE(pend.dx, rx); SEL_STORE(pend.dy, ry); SEL_STORE(pend.dz, rz);
                                SEL_STORE(pend.tx, ux); SEL_STORE(pend.ty, uy); SEL_STORE(pend.tz, uz);
                                SEL_STORE(pend.ior, n2); SEL_STORE(pend.depth, depth + 1);
                            }
                        }
                        np++;
                    } else if (valid) CNT(EV_DROP);
                }
                CNT(EV_REFL_EVAL);
                float Rt = m_refl + (1.0f - m_refl) * F;
                if (Rt > 0.0f) {
                    if (np < kp.stack_size) {
                        // the reflection ray is the next one visited: continue with it directly
                        CNT(EV_REFL_PUSH);
                        WCNT(WP_REFL);
                        float two_dni = dni * 2.0f;
                        ox = hx + onx * 0.001f; oy = hy + ony * 0.001f; oz = hz + onz * 0.001f;
                        dx = pdx - two_dni * onx; dy = pdy - two_dni * ony; dz = pdz - two_dni * onz;
                        thx = thx * Rt; thy = thy * Rt; thz = thz * Rt;
                        depth = depth + 1;                       // medium (cur_ior) unchanged (:639)
                        pop = false;
                    } else CNT(EV_DROP);
                }
            }
        }

        if (pop) {
            WCNT(WP_POP);
            if (np > 0) {
                WCNT(WP_POP_LOAD);
                np--;
                const int slot = np;
                if (PEND == PEND_HYBRID && slot >= RT_HYBRID_LDS_SLOTS) {
                    const float *q = pend_glb + slot * (11 * 256);
                    ox = q[0 * 256]; oy = q[1 * 256]; oz = q[2 * 256];
                    dx = q[3 * 256]; dy = q[4 * 256]; dz = q[5 * 256];
                    thx = q[6 * 256]; thy = q[7 * 256]; thz = q[8 * 256];
                    cur_ior = q[9 * 256]; depth = __float_as_int(q[10 * 256]);
                } else if (MEMP) {
                    const float *q = pend_mem + slot * (11 * 256);
                    ox = q[0 * 256]; oy = q[1 * 256]; oz = q[2 * 256];
                    dx = q[3 * 256]; dy = q[4 * 256]; dz = q[5 * 256];
                    thx = q[6 * 256]; thy = q[7 * 256]; thz = q[8 * 256];
                    cur_ior = q[9 * 256]; depth = __float_as_int(q[10 * 256]);
                } else {
#pragma unroll
                    for (int k = 0; k < (MEMP ? 1 : CAP); ++k) {
                        SEL_LOAD(ox, pend.ox); SEL_LOAD(oy, pend.oy); SEL_LOAD(oz, pend.oz);
                        SEL_LOAD(dx, pend.dx); SEL_LOAD(dy, pend.dy); SEL_LOAD(dz, pend.dz);
                        SEL_LOAD(thx, pend.tx); SEL_LOAD(thy, pend.ty); SEL_LOAD(thz, pend.tz);
                        SEL_LOAD(cur_ior, pend.ior); SEL_LOAD(depth, pend.depth);
                    }
                }
            } else {
                // path finished: next stratified sample of this pixel, or done
                WCNT(WP_FINISH);
                if (DEBUG && sample == 0) cap_nseg = seg;
                if (MULTI) {
                    accr = accr + fr; accg = accg + fg; accb = accb + fb;
                    sample++;
                    si++;
                    if (si == N) { si = 0; sj++; }
                    if (sample < nsamples) { if (SYNC) parked = true; else start_sample(); }
                    else active = false;
                } else {
                    active = false;
                }
                if (REFILL && !active) write_pixel();
            }
        }
        } while (false);
    }

    // ---- framebuffer write
    if (!DEBUG) {
        if (!REFILL && in_frame) write_pixel();
    } else {
        if (in_frame && cap_out) {
            size_t o = ((size_t)packed_row * kp.w + px) * 4;
            cap_out[o] = cap_id; cap_out[o + 1] = cap_t; cap_out[o + 2] = cap_depth; cap_out[o + 3] = (float)cap_nseg;
        }
        __syncthreads();
        if (ev_out && threadIdx.x < EV_COUNT) atomicAdd(&ev_out[threadIdx.x], (unsigned long long)cnt.lds_ev[threadIdx.x]);
    }
}

__global__ void reset_counters_kernel(unsigned int *counters) { counters[threadIdx.x * RT_SUBQUEUE_STRIDE] = 0u; }

// ------------------------------------------------------------------ launchers
template <int CAP, bool DEBUG, int PEND, bool MULTI, bool HIER, bool REFILL>
static int launch_variant(const KParams &kp, const float *d_scene, void *d_out, unsigned long long *d_ev,
                          float *d_cap, hipStream_t stream, LaunchInfo *info)
{
    constexpr bool LDSP = PEND == PEND_LDS || PEND == PEND_HYBRID;
    static_assert(!(LDSP && HIER), "the LDS pending slots and the candidate queue share one region");
    static_assert((PEND != PEND_GLOBAL && PEND != PEND_HYBRID) || REFILL,
                  "the global pending-ray workspace is sliced by persistent workgroup");
    if ((PEND == PEND_GLOBAL || PEND == PEND_HYBRID) && !kp.pend_ws) return (int)hipErrorInvalidValue;
    static_assert(!REFILL || !DEBUG, "the instrumented kernel keeps one pixel per lane");
    auto kfn = rt_kernel<CAP, DEBUG, PEND, MULTI, HIER, REFILL>;
    size_t lds = (size_t)(kp.scene.total_dwords - (HIER ? kp.scene.off_top : 0)) * 4 + (DEBUG ? EV_COUNT * 4 : 0);
    if (LDSP) lds = ((lds + 15) & ~(size_t)15) + (size_t)(PEND == PEND_HYBRID ? RT_HYBRID_LDS_SLOTS : CAP) * 11 * 256 * 4;
    if (HIER) lds = ((lds + 15) & ~(size_t)15) + (size_t)RT_QUEUE * 256 * 2;
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int bands = (kp.h + 15) / 16;
    int my_bands = (bands - kp.part + kp.nparts - 1) / kp.nparts;
    if (my_bands <= 0 || kp.w <= 0) return 0;
    dim3 grid((kp.w + 15) / 16, my_bands);
    if (REFILL) {
        // persistent waves: no more workgroups than the queue has work for or the GPU can hold
        if (!kp.work_counter) return (int)hipErrorInvalidValue;
        const int ntiles = ((kp.w + 7) / 8) * my_bands * 2;
        int wgs = (ntiles + 3) / 4;
        if (wgs > kp.resident_workgroups) wgs = kp.resident_workgroups;
        grid = dim3(wgs < 1 ? 1 : wgs, 1);
        hipLaunchKernelGGL(reset_counters_kernel, dim3(1), dim3(RT_SUBQUEUES), 0, stream, kp.work_counter);
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, stream, kp, d_scene, d_out, d_ev, d_cap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (info) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kfn)) == hipSuccess) info->vgprs = fa.numRegs;
        info->lds_bytes = (int)lds;
        info->pending_cap = CAP;
    }
    return 0;
}

static int needed_cap(const KParams &kp)
{
    int need = kp.stack_size < kp.max_depth - 1 ? kp.stack_size : kp.max_depth - 1;
    return need < 1 ? 1 : need;
}

int launch_render(const KParams &kp, const float *d_scene, void *d_out, void *stream, LaunchInfo *info)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int need = needed_cap(kp);
    static const bool no_ldsp = getenv("RT_NO_LDSP") != nullptr;
    const bool multi = kp.spp > 1;
#define RT_LAUNCH(CAP_, LDSP_)                                                                              \
    (multi ? launch_variant<CAP_, false, (LDSP_) ? PEND_LDS : PEND_REG, true, false, false>(kp, d_scene, d_out, nullptr, nullptr, s, info) \
           : launch_variant<CAP_, false, (LDSP_) ? PEND_LDS : PEND_REG, false, false, false>(kp, d_scene, d_out, nullptr, nullptr, s, info))
#define RT_LAUNCH_HIER(CAP_, PEND_)                                                                         \
    (multi ? launch_variant<CAP_, false, PEND_, true, true, true>(kp, d_scene, d_out, nullptr, nullptr, s, info) \
           : launch_variant<CAP_, false, PEND_, false, true, true>(kp, d_scene, d_out, nullptr, nullptr, s, info))
    // Large scenes run as persistent waves with pixel refill (measured 1.1-2.4x there: 64 tori
    // 4K 2.43 -> 2.30 ms, 1024 objects 4K depth 4 13.4 -> 10.6 ms, 256 spheres 4K depth 16
    // 2250 -> 928 ms).  The 5-object scene keeps one pixel per lane: there the lanes' loss of
    // lock-step costs more than the idle lanes (depth 8 / stack 8: 0.96 -> 1.53 ms).
    if (kp.use_hierarchy && kp.scene.n1 > 0) {
        // pending rays in the global workspace: 123 VGPRs whatever the stack size (register slots:
        // 147 / 168 / 256 for 3 / 7 / 16), measured 1.06-1.46x faster than register slots
        return RT_LAUNCH_HIER(RT_PEND_MAX, PEND_GLOBAL);
    }
    if ((size_t)kp.scene.total_dwords * 4 + 3 * 11 * 256 * 4 > 150 * 1024) {
        // per-object loop forced on a scene whose blob leaves no LDS for pending slots
        // (RT_FLAG_NO_ACCEL with ~1000 tori): all slots in the workspace
        KParams kw = kp;
        kw.whole_tiles = 1;
        return multi ? launch_variant<RT_PEND_MAX, false, PEND_GLOBAL, true, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info)
                     : launch_variant<RT_PEND_MAX, false, PEND_GLOBAL, false, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info);
    }
    if (need > 3) {
        // Small scene, more than three pending rays possible: persistent waves that take whole
        // tiles (lanes stay in lock-step), slots 0-2 in LDS and the deeper ones in the global
        // workspace.  Keeps the 34 KB / 4-waves-per-SIMD footprint of the three-slot kernel
        // (default scene 4K: depth 8 stack 8 0.60 ms, against 0.91 with seven LDS slots at two
        // workgroups per CU and 0.96 with seven register slots; depth 16 stack 16 2.8 ms
        // against 8.9 with sixteen register slots).
        KParams kw = kp;
        kw.whole_tiles = 1;
        return multi ? launch_variant<RT_PEND_MAX, false, PEND_HYBRID, true, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info)
                     : launch_variant<RT_PEND_MAX, false, PEND_HYBRID, false, false, true>(kw, d_scene, d_out, nullptr, nullptr, s, info);
    }
    // 3 slots x 11 dwords x 256 threads = 33 KB of LDS per workgroup: fits 4 workgroups per CU
    // next to a small scene, and frees the registers for a fourth wave per SIMD
    if (!no_ldsp) return RT_LAUNCH(3, true);
    return RT_LAUNCH(3, false);
#undef RT_LAUNCH
#undef RT_LAUNCH_HIER
}

int launch_debug(const KParams &kp, const float *d_scene, unsigned long long *d_events, float *d_capture,
                 void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return launch_variant<16, true, PEND_REG, true, false, false>(kp, d_scene, nullptr, d_events, d_capture, s, nullptr);
}

#ifndef RT_FAST
// Exhaustive comparison of the range-restricted sqrt / reciprocal / division cores with the
// compiler's full expansions.  counts[0]: sqrt over all x in [2^-96, 2^96); counts[1]:
// 1/sqrt over the same x; counts[2]: 1/s over all s in [2^-48, 2^48]; counts[3]: log2's
// (m - 1)/(m + 1) over all m in [1, 2).  A count is the number of operands whose results
// differ in any bit.  counts[4], counts[5]: x / 3 and x / 27 over all x of both signs with |x|
// in [2^-96, 2^96).
__global__ __launch_bounds__(256) void rt_selftest_kernel(unsigned long long *counts)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t stride = gridDim.x * 256u;
    unsigned int bad0 = 0, bad1 = 0, bad2 = 0, bad3 = 0;
    for (uint32_t u = 0x0f800000u + tid; u < 0x6f800000u; u += stride) {
        float x = bits_f(u);
        bad0 += f_bits(sqrt_core(x)) != f_bits(__builtin_sqrtf(x));
        bad1 += f_bits(rcp_core(sqrt_core(x))) != f_bits(1.0f / __builtin_sqrtf(x));
        if (u - stride < 0x0f800000u) {      // first trip of this thread: the guards themselves
            bad0 += !sqrt_in_core_range(x);
            bad0 += sqrt_in_core_range(bits_f(0x0f7fffffu)) + sqrt_in_core_range(bits_f(0x6f800000u)) +
                    sqrt_in_core_range(0.0f) + sqrt_in_core_range(-1.0f) + sqrt_in_core_range(bits_f(0x7f800000u)) +
                    sqrt_in_core_range(bits_f(0x7fc00000u)) + sqrt_in_core_range(bits_f(0x00000001u));
        }
    }
    for (uint32_t u = 0x27800000u + tid; u <= 0x57800000u; u += stride) {     // 2^-48 .. 2^48
        float sv = bits_f(u);
        bad2 += f_bits(rcp_core(sv)) != f_bits(1.0f / sv);
    }
    for (uint32_t u = 0x3f800000u + tid; u < 0x40000000u; u += stride) {
        float m = bits_f(u);
        bad3 += f_bits(div_core(m - 1.0f, m + 1.0f)) != f_bits((m - 1.0f) / (m + 1.0f));
    }
    unsigned int bad4 = 0, bad5 = 0;
    for (uint32_t u = 0x0f800000u + tid; u < 0x6f800000u; u += stride) {
        for (uint32_t sign = 0; sign < 2; ++sign) {
            float x = bits_f(u | (sign << 31));
            bad4 += f_bits(div_const_core<3>(x)) != f_bits(x / 3.0f);
            bad5 += f_bits(div_const_core<27>(x)) != f_bits(x / 27.0f);
            bad4 += !div_const_in_core_range(x);
        }
    }
    if (tid == 0)
        bad4 += div_const_in_core_range(0.0f) + div_const_in_core_range(-0.0f) + div_const_in_core_range(bits_f(0x0f7fffffu)) +
                div_const_in_core_range(bits_f(0xef800000u)) + div_const_in_core_range(bits_f(0x7f800000u)) +
                div_const_in_core_range(bits_f(0xffc00000u)) + div_const_in_core_range(bits_f(0x80000001u));
    if (bad4) atomicAdd(&counts[4], (unsigned long long)bad4);
    if (bad5) atomicAdd(&counts[5], (unsigned long long)bad5);
    if (bad0) atomicAdd(&counts[0], (unsigned long long)bad0);
    if (bad1) atomicAdd(&counts[1], (unsigned long long)bad1);
    if (bad2) atomicAdd(&counts[2], (unsigned long long)bad2);
    if (bad3) atomicAdd(&counts[3], (unsigned long long)bad3);
}

int launch_selftest(unsigned long long *d_counts, void *stream)
{
    hipLaunchKernelGGL(rt_selftest_kernel, dim3(256 * 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_counts);
    return (int)hipGetLastError();
}
#endif

}  // namespace RT_NS
