/*
 * rt_oracle.c -- CPU restatement of the reference renderer's per-pixel path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the oracle the HIP renderer is checked against and
 * the host-CPU baseline that bench.py times.  The shipped renderer never links, loads
 * or calls it.
 *
 * What it restates: /root/reference/fragment_shader.glsl, function by function, in the
 * shader's own structure (explicit LIFO ray stack, per-object loop, Ferrari quartic).
 * Each function cites the shader lines it follows.
 *
 * Numerics.  The reference has no CPU path of its own; the only place its shader can
 * run without a GPU is Mesa llvmpipe, and that is what the golden vectors in
 * tests/golden come from.  To be comparable bit for bit, every expression below is
 * written in the operation order of the code llvmpipe actually executes for that shader
 * (its final NIR, obtained with LP_DEBUG=fs, Mesa 23.2.1 / LLVM 15):
 *   - binary32 everywhere, no FMA contraction of a*b+c (build with -ffp-contract=off),
 *   - 3-component dot products are summed (z*z' + y*y') + x*x',
 *   - normalize(v) = v * (1 / sqrt(dot(v,v))), '/' and sqrt correctly rounded,
 *   - denormals are flushed (FTZ + DAZ, set per thread),
 *   - pow = exp2(y * log2 x), cos, acos are llvmpipe's own polynomial kernels; those DO
 *     use fused multiply-add (llvm.fmuladd -> vfmadd on the machine that produced the
 *     golden vectors), restated here with fmaf,
 *   - a handful of algebraic rewrites the GLSL compiler made (noted where they occur).
 * Pinned by tests/test_oracle_golden.py against llvmpipe renders of the reference shader.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_STACK 64
#define ORC_MAX_LIGHT_GRID 8

/* Mirrors struct SceneObject, fragment_shader.glsl:41-56 (16 dwords, 64 bytes). */
typedef struct {
    int32_t type;            /* 1 sphere, 2 torus (fragment_shader.glsl:19-20) */
    float center[3];
    float radius;
    float normal[3];
    float major_radius, minor_radius;
    float color[3];
    float reflectivity, transparency, refractive_index;
} orc_object;

/* Uniforms of fragment_shader.glsl:12-15 plus the constants the shader hard-codes. */
typedef struct {
    float cam_to_world[16];  /* row-major, rows [x|y|view|eye] as main.py:269-274 builds it */
    float focal;             /* u_focal_length */
    float light[3];          /* u_light_pos */
    int32_t w, h;            /* u_resolution */
    int32_t max_depth;       /* fragment_shader.glsl:653 (4) */
    int32_t stack_size;      /* fragment_shader.glsl:547 (3) */
    int32_t shadow_hops;     /* fragment_shader.glsl:453 (2) */
    int32_t spp_grid;        /* N: N x N stratified samples per pixel; 1 = reference */
    int32_t light_grid;      /* G: G x G samples of a square area light; 0 or 1 = the reference's point light */
    float light_size;        /* edge of that square (horizontal, centred on `light`) */
    uint32_t flags;          /* ORC_FLAG_* */
} orc_params;
#define ORC_FLAG_NEWTON_POLISH 1u   /* not the reference: polish torus roots, drop roots that are not on the torus */

/* ---- work counters (events), used for the algorithmic FLOP total ---- */
enum {
    EV_PIXEL_PATH, EV_TRACE, EV_BOUND_CULL_DISC, EV_BOUND_TBOUND, EV_BOUND_TORUS_RADIUS,
    EV_SPHERE_CALL, EV_SPHERE_ROOT, EV_SPHERE_HIT,
    EV_PLANE_CALL, EV_PLANE_T, EV_PLANE_HIT,
    EV_TORUS_CALL, EV_QUAD_FULL, EV_QUAD_EARLY, EV_BIQ_Z, EV_ROOT_OUT,
    EV_CUBIC, EV_CARDANO, EV_TRIG, EV_FERRARI_TAIL, EV_TORUS_HIT,
    EV_SEG_MISS, EV_SEG_HIT, EV_FRESNEL_TIR, EV_FRESNEL_FULL, EV_LOCAL,
    EV_PHONG_SHADOWED, EV_PHONG_FULL, EV_ATTEN_FIXED, EV_HOP_HIT_TEST, EV_HOP_FILTER, EV_HOP_CONTINUE,
    EV_REFR_TRY, EV_REFR_PUSH, EV_REFL_EVAL, EV_REFL_PUSH, EV_DROP,
    EV_COUNT
};

/* FLOPs per event: every scalar add/sub/mul/div/sqrt/rsqrt/floor/min/max/mod/pow/cos/acos
 * in the reference GLSL as written is 1; compare/negate/abs/sign/select are 0; dot3 = 5,
 * cross = 9, normalize = 9, length = 6, reflect = 12, refract = 22, mix(vec3) = 9, clamp = 2. */
static const int ev_flops[EV_COUNT] = {
    [EV_PIXEL_PATH] = 30,       /* main(): uv 6, direction 15, normalize 9            :652-668 */
    [EV_TRACE] = 0,
    [EV_BOUND_CULL_DISC] = 17,  /* oc 3, b 5, c 7, disc 2                             :399-402 */
    [EV_BOUND_TBOUND] = 19,     /* + t_bound 2                                        :410     */
    [EV_BOUND_TORUS_RADIUS] = 1,/* major + minor                                      :395     */
    [EV_SPHERE_CALL] = 25,      /* oc 3, a 5, b 6, c 7, disc 4                        :77-81   */
    [EV_SPHERE_ROOT] = 4,       /* (-b -+ sqrt(disc)) / (2a)                          :87,:89  */
    [EV_SPHERE_HIT] = 18,       /* position 6, normal 12                              :95-96   */
    [EV_PLANE_CALL] = 5,        /* dot                                                :116     */
    [EV_PLANE_T] = 2,           /* (ground - o.y) / den                               :118     */
    [EV_PLANE_HIT] = 10,        /* position 6, floor floor add mod                    :124,:137*/
    [EV_TORUS_CALL] = 131,      /* frame 36, lo 3, ro/rd 30, m n 10, partial dots 9, k 4,
                                   A 1, B 8, C 6, D 5 (=112) + quartic depress 19     :314-339,:252-256 */
    [EV_QUAD_FULL] = 11,        /* solve_quadratic, both roots                        :166-177 */
    [EV_QUAD_EARLY] = 4,        /* discriminant only                                  :166-170 */
    [EV_BIQ_Z] = 2,             /* sqrt(max(0,z))                                     :268     */
    [EV_ROOT_OUT] = 1,          /* root - offset                                      :269,:271,:298,:302 */
    [EV_CUBIC] = 26,            /* resolvent coefficients 5, solve_cubic common 21    :281,:201-215 */
    [EV_CARDANO] = 10,          /*                                                    :218-225 */
    [EV_TRIG] = 23,             /*                                                    :229-235 */
    [EV_FERRARI_TAIL] = 9,      /* w 1, term_A 3, term_B 3, A-B 1, A+B 1              :290-301 */
    [EV_TORUS_HIT] = 54,        /* position 6, local 6, alpha 5, scale 4, normalize 9,
                                   to world 15, normalize 9                           :359-367 */
    [EV_SEG_MISS] = 17,         /* t 2, mix 9, accumulate 6                           :575-578 */
    [EV_SEG_HIT] = 12,          /* dot 5, cos 5, local_coef 2                         :585,:596,:600 */
    [EV_FRESNEL_TIR] = 9,       /*                                                    :525-531 */
    [EV_FRESNEL_FULL] = 14,     /*                                                    :525-533 */
    [EV_LOCAL] = 9,             /* final += local * coef * throughput                 :603     */
    [EV_PHONG_SHADOWED] = 8,    /* ambient 3, dot 5                                   :490-497 */
    [EV_PHONG_FULL] = 73,       /*                                                    :490-513 */
    [EV_ATTEN_FIXED] = 39,      /* length 9, origin 18, direction 12                  :443-448 */
    [EV_HOP_HIT_TEST] = 1,      /* hit.t + distance_traveled                          :457     */
    [EV_HOP_FILTER] = 11,       /* object_filter 3, *= 3, dot 5                       :462-468 */
    [EV_HOP_CONTINUE] = 7,      /* distance 1, origin 6                               :473-474 */
    [EV_REFR_TRY] = 28,         /* n1/n2 1, refract 22, dot 5                         :613-615 */
    [EV_REFR_PUSH] = 16,        /* origin 6, throughput 10                            :618-620 */
    [EV_REFL_EVAL] = 3,         /* total_reflectivity                                 :629     */
    [EV_REFL_PUSH] = 21,        /* origin 6, reflect 12, throughput 3                 :635-637 */
    [EV_DROP] = 0,
};

typedef struct { uint64_t c[EV_COUNT]; } counters;
#define CNT(ev) do { if (cnt) cnt->c[ev]++; } while (0)

/* ---- small vector helpers in llvmpipe's operation order ---- */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* dot(a,b) as llvmpipe sums it: (z + y) + x */
static inline float dot3(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
static inline v3 cross3(v3 a, v3 b)
{ return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float rsq(float x) { return 1.0f / sqrtf(x); }
static inline v3 normalize3(v3 a) { return vscale(a, rsq(dot3(a, a))); }

static inline float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
/* x86 MAXPS/MINPS semantics as llvmpipe uses them: the result is the second operand
 * unless the comparison holds, so a NaN in either place yields the second operand. */
static inline float x86_max(float a, float b) { return a > b ? a : b; }
static inline float x86_min(float a, float b) { return a < b ? a : b; }
static inline float sign_f(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* ---- llvmpipe's transcendental kernels ---- */

/* log2(x): mantissa/exponent split, y=(m-1)/(m+1), degree-4 polynomial in y^2 (even and
 * odd halves evaluated separately), all multiply-adds fused. */
static float lp_log2(float x)
{
    uint32_t i = f_bits(x);
    float e = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
    float m = bits_f((i & 0x007fffffu) | 0x3f800000u);
    float y = (m - 1.0f) / (m + 1.0f);
    float z = y * y;
    float z2 = z * z;
    const float c0 = 0x1.715476p+1f;   /* 2.88539   */
    const float c1 = 0x1.ec6ff2p-1f;   /* 0.961792  */
    const float c2 = 0x1.27a642p-1f;   /* 0.577440  */
    const float c3 = 0x1.9d062cp-2f;   /* 0.403344  */
    const float c4 = 0x1.a07ab2p-2f;   /* 0.406718  */
    float even = fmaf(z2, fmaf(z2, c4, c2), c0);
    float odd = fmaf(z2, c3, c1);
    float p = fmaf(odd, z, even);
    float r = fmaf(y, p, e);
    if (x >= INFINITY) r = INFINITY;
    if (x == 0.0f) r = -INFINITY;
    if (!(x >= 0.0f)) r = NAN;   /* x < 0 or NaN */
    return r;
}

/* exp2(t) for t already clamped: floor/fraction split, degree-5 polynomial on the fraction. */
static float lp_exp2_clamped(float t)
{
    float fl = floorf(t);
    float fpart = t - fl;
    int32_t ipart = (int32_t)fl;
    float scale = bits_f((uint32_t)(ipart + 127) << 23);
    float x2 = fpart * fpart;
    const float d1 = 0x1.62e4f6p-1f;   /* 0.693153   */
    const float d2 = 0x1.ebd5a8p-3f;   /* 0.240154   */
    const float d3 = 0x1.c95446p-5f;   /* 0.0558263  */
    const float d4 = 0x1.26900cp-7f;   /* 0.00898934 */
    const float d5 = 0x1.ec320ap-10f;  /* 0.00187758 */
    float odd = fmaf(x2, fmaf(x2, d5, d3), d1);
    float even = fmaf(x2, fmaf(x2, d4, d2), 1.0f);
    float p = fmaf(odd, fpart, even);
    return scale * p;
}

float orc_pow(float x, float y)
{
    if (!(x < 0.0f || x > 0.0f)) return 0.0f;      /* x == 0 (or NaN) -> 0 */
    float t = lp_log2(x) * y;
    if (t != t) return NAN;
    t = x86_min(128.0f, t);
    t = x86_max(-0x1.fbfffep+6f, t);               /* -126.99999 */
    return lp_exp2_clamped(t);
}

/* cos(x): Cephes-style octant reduction with a three-part pi/4, fused steps. */
float orc_cos(float x)
{
    float xa = fabsf(x);
    float sy = xa * 0x1.45f306p+0f;                /* 4/pi */
    int32_t j = (int32_t)sy;
    j = (j + 1) & ~1;
    float yf = (float)j;
    int32_t emm2 = j - 2;
    uint32_t sign_bit = ((uint32_t)(~emm2) & 4u) << 29;
    int use_sin_poly = (emm2 & 2) == 0;
    float r = fmaf(yf, -0.78515625f, xa);
    r = fmaf(yf, -0x1.fb4p-13f, r);                /* -2.4187564849853515625e-4 */
    r = fmaf(yf, -0x1.4442d2p-25f, r);             /* -3.77489497744594108e-8  */
    float z = r * r;
    float res;
    if (use_sin_poly) {
        float s = fmaf(z, -0x1.9943f2p-13f, 0x1.11073cp-7f);
        s = fmaf(s, z, -0x1.555546p-3f);
        s = s * z;
        res = fmaf(s, r, r);
    } else {
        float c = fmaf(z, 0x1.99eb9cp-16f, -0x1.6c0c34p-10f);
        c = fmaf(c, z, 0x1.55554ap-5f);
        c = c * z;
        c = c * z;
        c = c - z * 0.5f;
        res = c + 1.0f;
    }
    res = bits_f(f_bits(res) ^ sign_bit);
    if ((f_bits(x) & 0x7f800000u) == 0x7f800000u) return NAN;
    res = x86_min(res, 1.0f);
    res = x86_max(res, -1.0f);
    return res;
}

/* acos(x) as the GLSL compiler lowers it: pi/2 - sign(x)*(pi/2 - sqrt(1-|x|)*P(|x|)). */
float orc_acos(float x)
{
    const float half_pi = 0x1.921fb6p+0f;
    float s = sign_f(x);
    float ax = fabsf(x);
    float t = sqrtf(1.0f - ax);
    float p = ax * -0x1.8334bep-6f;     /* -0.023633 */
    p = 0x1.4d1b0ep-4f + p;             /*  0.081325 */
    p = ax * p;
    p = -0x1.b78128p-3f + p;            /* -0.214602 */
    p = ax * p;
    p = half_pi + p;
    float q = t * p;
    q = half_pi - q;
    q = s * q;
    return half_pi - q;
}

/* ---- ray / hit records (fragment_shader.glsl:23-38, :60-65) ---- */
typedef struct { v3 o, d; } ray_t;
typedef struct {
    int hit; float t; v3 pos, n, color; float refl, transp, ior; int id;
} hit_t;
typedef struct { ray_t ray; v3 thr; int depth; float ior; } raystate_t;

typedef struct {
    const orc_object *objs; int n;
    v3 light, cam_pos;
    int max_depth, stack_size, hops;
    int light_grid;                         /* > 1: area light, see phong() */
    float light_off[ORC_MAX_LIGHT_GRID];    /* offset of sample column/row i from the light's centre */
    float inv_light;                        /* 1 / (light_grid * light_grid) */
    int polish;                             /* ORC_FLAG_NEWTON_POLISH */
} scene_ctx;

/* intersect_sphere, fragment_shader.glsl:74-103 */
static int intersect_sphere(ray_t r, const orc_object *s, hit_t *h, counters *cnt)
{
    CNT(EV_SPHERE_CALL);
    v3 c = V(s->center[0], s->center[1], s->center[2]);
    v3 oc = vsub(r.o, c);
    float a = dot3(r.d, r.d);
    float b = 2.0f * dot3(oc, r.d);
    float cc = dot3(oc, oc) - s->radius * s->radius;
    float disc = b * b - (4.0f * a) * cc;
    if (disc < 0.0f) return 0;
    float sq = sqrtf(disc);
    float two_a = 2.0f * a;
    CNT(EV_SPHERE_ROOT);
    float t = (-b - sq) / two_a;
    if (t < 0.0f) { CNT(EV_SPHERE_ROOT); t = (-b + sq) / two_a; }
    if (!(t > 0.001f)) return 0;
    CNT(EV_SPHERE_HIT);
    h->hit = 1; h->t = t;
    h->pos = vadd(r.o, vscale(r.d, t));
    h->n = normalize3(vsub(h->pos, c));
    h->color = V(s->color[0], s->color[1], s->color[2]);
    h->refl = s->reflectivity; h->transp = s->transparency; h->ior = s->refractive_index;
    return 1;
}

/* intersect_plane, fragment_shader.glsl:107-146.  dot(d,(0,1,0)) is d.y; mod(s,2)==0 is
 * evaluated as s == 2*floor(s/2). */
static int intersect_plane(ray_t r, hit_t *h, counters *cnt)
{
    CNT(EV_PLANE_CALL);
    float den = r.d.y;
    int ok = fabsf(den) > 0.0001f;
    if (!ok) return 0;
    CNT(EV_PLANE_T);
    float t = (-1.0f - r.o.y) / den;
    if (!(t > 0.001f)) return 0;
    CNT(EV_PLANE_HIT);
    h->hit = 1; h->t = t;
    h->pos = vadd(r.o, vscale(r.d, t));
    h->n = V(0.0f, 1.0f, 0.0f);
    h->transp = 0.0f; h->ior = 1.0f; h->refl = 0.05f;
    float s = floorf(h->pos.x) + floorf(h->pos.z);
    float two_fl = 2.0f * floorf(s / 2.0f);
    float col = (s == two_fl) ? 0.9f : 0.1f;
    h->color = V(col, col, col);
    return 1;
}

#define Q_EPS 1e-7f   /* fragment_shader.glsl:149 */

/* solve_quadratic (fragment_shader.glsl:159-180) for the monic calls solve_quartic makes:
 * a = 1, so inv_2a = 0.5 and 4*a*c = 4*c.  bb is b*b as the compiled shader has it. */
static int quad_monic(float bb, float b, float c, float roots[2], counters *cnt)
{
    float disc = bb - 4.0f * c;
    if (disc < -Q_EPS) { CNT(EV_QUAD_EARLY); return 0; }
    CNT(EV_QUAD_FULL);
    disc = fmaxf(0.0f, disc);
    float sq = sqrtf(disc);
    roots[0] = (-b + sq) * 0.5f;
    roots[1] = (-b - sq) * 0.5f;
    return disc < Q_EPS ? 1 : 2;
}

/* solve_cubic (fragment_shader.glsl:192-238) as called from solve_quartic with a = 1:
 * returns roots[0] only (the only one the caller reads, :285). */
static float cubic_monic_root0(float an, float bn, float cn, counters *cnt)
{
    CNT(EV_CUBIC);
    float an2 = an * an;
    float p = bn - an2 / 3.0f;
    float offset = an / 3.0f;
    float q = (cn - (an * bn) / 3.0f) + ((2.0f * an2) * an) / 27.0f;
    float half_q = 0.5f * q;
    float p3 = (p * p) * p;
    float disc = half_q * half_q + p3 / 27.0f;
    if (disc >= -Q_EPS) {
        CNT(EV_CARDANO);
        disc = fmaxf(0.0f, disc);
        float sq = sqrtf(disc);
        float A = -half_q + sq;
        float B = -half_q - sq;
        float rootA = sign_f(A) * orc_pow(fabsf(A), 0x1.555556p-2f);
        float rootB = sign_f(B) * orc_pow(fabsf(B), 0x1.555556p-2f);
        return (rootA - offset) + rootB;       /* compiled order of rootA + rootB - offset */
    }
    CNT(EV_TRIG);
    float rho = sqrtf(-p3 / 27.0f);
    float x = -half_q / rho;
    x = fmaxf(x, -1.0f);                       /* NaN -> -1 */
    x = fminf(x, 1.0f);
    float theta = orc_acos(x) / 3.0f;
    float m = 2.0f * sqrtf(-p / 3.0f);
    return m * orc_cos(theta) - offset;
}

/* solve_quartic, fragment_shader.glsl:250-305 (monic; Ferrari). */
static int solve_quartic(float a, float b, float c, float d, float roots[4], counters *cnt)
{
    float a2 = a * a;
    float p = b - 0.375f * a2;
    float q = (c - (0.5f * a) * b) + (0.125f * a2) * a;
    float r = ((d - (0.25f * a) * c) + (0.0625f * a2) * b) - (0.01171875f * a2) * a2;
    float offset = 0.25f * a;
    int n = 0;
    if (fabsf(q) < Q_EPS) {                    /* biquadratic, :262-276 */
        float y2[2];
        int ny = quad_monic(p * p, p, r, y2, cnt);
        for (int i = 0; i < ny; ++i) {
            float z = y2[i];
            if (z >= -Q_EPS) {
                CNT(EV_BIQ_Z);
                float y = sqrtf(fmaxf(0.0f, z));
                CNT(EV_ROOT_OUT);
                roots[n++] = y - offset;
                if (y > Q_EPS) { CNT(EV_ROOT_OUT); roots[n++] = -y - offset; }
            }
        }
        return n;
    }
    float u = cubic_monic_root0(2.0f * p, p * p - 4.0f * r, -(q * q), cnt);
    if (u < 0.0f) return 0;
    CNT(EV_FERRARI_TAIL);
    float w = sqrtf(u);
    float term_A = 0.5f * (p + u);             /* compiled form of 0.5*p + 0.5*u */
    float term_B = (0.5f * q) / (w + Q_EPS);
    float qr[2];
    float bb = fabsf(u);                       /* compiled form of w*w */
    int n1 = quad_monic(bb, w, term_A - term_B, qr, cnt);
    for (int i = 0; i < n1; ++i) { CNT(EV_ROOT_OUT); roots[n++] = qr[i] - offset; }
    int n2 = quad_monic(bb, -w, term_A + term_B, qr, cnt);
    for (int i = 0; i < n2; ++i) { CNT(EV_ROOT_OUT); roots[n++] = qr[i] - offset; }
    return n;
}

/* Entry point for known-answer tests of the solver alone. */
int orc_solve_quartic(float a, float b, float c, float d, float *roots)
{
    _mm_setcsr(_mm_getcsr() | 0x8040);
    return solve_quartic(a, b, c, d, roots, NULL);
}

/*
 * Newton polish of a torus root (NOT part of the reference, whose solve ends with Ferrari's roots,
 * fragment_shader.glsl:345-352; optional, ORC_FLAG_NEWTON_POLISH).  The quartic's coefficients are
 * differences of terms of size |ro|^4, so in binary32 its roots are off by 1e-4..1e-2 of t and some
 * are not intersections at all.  The polish leaves the polynomial and works on the torus itself,
 * in its own frame: g(p) = (rho - R)^2 + y^2 - r^2 with rho = sqrt(x^2 + z^2) is zero on the
 * surface and well conditioned near it.  Five Newton steps on g(ro + t rd), each at most r long;
 * the best of the six points visited replaces the root if it is past 0.001 and within 4r of
 * the start (it cannot wander to another crossing of the tube), otherwise the root stays.  A root
 * whose best |g| is still above 0.01 r^2 does not lie on the torus: the hit is dropped.
 * Returns 1 with *t updated, or 0 for "no intersection".  Plain binary32 operations in this order;
 * the HIP kernel (rt_intersect.h: torus_polish) runs the same sequence.
 */
static float torus_g(v3 ro, v3 rd, float R, float r2, float t, float *dg)
{
    float px = ro.x + t * rd.x, py = ro.y + t * rd.y, pz = ro.z + t * rd.z;
    float rho = sqrtf(px * px + pz * pz);
    float a = rho - R;
    float g = (a * a + py * py) - r2;
    *dg = 2.0f * (a * ((px * rd.x + pz * rd.z) / rho) + py * rd.y);
    return g;
}
#ifndef ORC_POLISH_STEPS
#define ORC_POLISH_STEPS 5
#define ORC_POLISH_WINDOW 4.0f
#endif
static int torus_polish(v3 ro, v3 rd, float R, float r, float *t_io)
{
    const float r2 = r * r, t0 = *t_io;
    float t = t0, dg;
    float g = torus_g(ro, rd, R, r2, t, &dg);
    float best_t = t0, best_g = fabsf(g);
    for (int it = 0; it < ORC_POLISH_STEPS; ++it) {
        if (!(fabsf(dg) > 0.0f)) break;                 /* zero or NaN slope */
        float step = g / dg;
        step = step > r ? r : step;
        step = step < -r ? -r : step;
        t = t - step;
        g = torus_g(ro, rd, R, r2, t, &dg);
        if (fabsf(g) < best_g) { best_g = fabsf(g); best_t = t; }
    }
    if (best_t > 0.001f && fabsf(best_t - t0) <= ORC_POLISH_WINDOW * r) *t_io = best_t;
    else best_g = fabsf(torus_g(ro, rd, R, r2, t0, &dg));
    return best_g <= 0.01f * r2;                        /* false for NaN */
}

/* intersect_torus, fragment_shader.glsl:309-375 */
static int intersect_torus(ray_t ray, const orc_object *s, hit_t *h, counters *cnt, int polish)
{
    CNT(EV_TORUS_CALL);
    v3 c = V(s->center[0], s->center[1], s->center[2]);
    v3 lo = vsub(ray.o, c);
    v3 w = normalize3(V(s->normal[0], s->normal[1], s->normal[2]));
    float sel = fabsf(w.y) > 0.99f ? 1.0f : 0.0f;
    v3 helper = V(sel, 1.0f - sel, 0.0f);      /* (1,0,0) or (0,1,0) */
    v3 u = normalize3(cross3(w, helper));
    v3 v = cross3(w, u);
    v3 ro = V(dot3(lo, u), dot3(lo, w), dot3(lo, v));
    v3 rd = V(dot3(ray.d, u), dot3(ray.d, w), dot3(ray.d, v));
    float R = s->major_radius, r = s->minor_radius;
    float m = dot3(ro, ro);
    float n = dot3(ro, rd);
    float rd_dxz = rd.x * rd.x + rd.z * rd.z;
    float ro_rd_dxz = ro.x * rd.x + ro.z * rd.z;
    float ro_dxz = ro.x * ro.x + ro.z * ro.z;
    float k = (m + R * R) - r * r;
    float four_n = 4.0f * n;
    float four_R2 = (4.0f * R) * R;
    float A = four_n;
    float B = (2.0f * k + four_n * n) - four_R2 * rd_dxz;
    float C = four_n * k - ((8.0f * R) * R) * ro_rd_dxz;
    float D = k * k - four_R2 * ro_dxz;
    float roots[4];
    int nr = solve_quartic(A, B, C, D, roots, cnt);
    float t = 1e20f; int found = 0;
    for (int i = 0; i < nr; ++i)
        if (roots[i] > 0.001f && roots[i] < t) { t = roots[i]; found = 1; }
    if (!found) return 0;
    if (polish && !torus_polish(ro, rd, R, r, &t)) return 0;
    CNT(EV_TORUS_HIT);
    h->hit = 1; h->t = t;
    h->pos = vadd(ray.o, vscale(ray.d, t));
    v3 lp = vadd(ro, vscale(rd, t));
    float alpha = R / sqrtf(lp.x * lp.x + lp.z * lp.z);
    float f = 1.0f - alpha;
    v3 nl = normalize3(V(lp.x * f, lp.y, lp.z * f));
    v3 nw = vadd(vadd(vscale(u, nl.x), vscale(w, nl.y)), vscale(v, nl.z));
    h->n = normalize3(nw);
    h->color = V(s->color[0], s->color[1], s->color[2]);
    h->refl = s->reflectivity; h->transp = s->transparency; h->ior = s->refractive_index;
    return 1;
}

/* trace, fragment_shader.glsl:379-435 */
static hit_t trace(const scene_ctx *sc, ray_t ray, counters *cnt)
{
    CNT(EV_TRACE);
    hit_t best; memset(&best, 0, sizeof best);
    best.t = 1e30f; best.id = -1;
    for (int i = 0; i < sc->n; ++i) {
        const orc_object *s = &sc->objs[i];
        float br;
        if (s->type == 1) br = s->radius;
        else { CNT(EV_BOUND_TORUS_RADIUS); br = s->major_radius + s->minor_radius; }
        v3 oc = vsub(ray.o, V(s->center[0], s->center[1], s->center[2]));
        float b = dot3(oc, ray.d);
        float c = dot3(oc, oc) - br * br;
        float disc = b * b - c;
        if (disc < 0.0f) { CNT(EV_BOUND_CULL_DISC); continue; }
        CNT(EV_BOUND_TBOUND);
        float t_bound = -b - sqrtf(disc);
        if (t_bound > best.t) continue;
        hit_t cur; cur.hit = 0;
        if (s->type == 1) intersect_sphere(ray, s, &cur, cnt);
        else intersect_torus(ray, s, &cur, cnt, sc->polish);
        if (cur.hit && cur.t < best.t) { best = cur; best.id = i; }
    }
    hit_t ph; ph.hit = 0;
    intersect_plane(ray, &ph, cnt);
    if (ph.hit && ph.t < best.t) { best = ph; best.id = sc->n; }
    return best;
}

/* calculate_light_attenuation, fragment_shader.glsl:439-483 */
static v3 light_attenuation(const scene_ctx *sc, v3 light, v3 point, counters *cnt)
{
    CNT(EV_ATTEN_FIXED);
    v3 filt = V(1.0f, 1.0f, 1.0f);
    v3 to_l = vsub(light, point);
    float d2 = dot3(to_l, to_l);
    float light_dist = sqrtf(d2);
    v3 L = vscale(to_l, rsq(d2));
    v3 step = vscale(L, 0.001f);
    ray_t sr; sr.o = vadd(point, step); sr.d = L;
    float travelled = 0.0f;
    for (int i = 0; i < sc->hops; ++i) {
        hit_t h = trace(sc, sr, cnt);
        if (!h.hit) break;
        CNT(EV_HOP_HIT_TEST);
        float reach = h.t + travelled;
        if (!(reach < light_dist)) break;
        CNT(EV_HOP_FILTER);
        filt.x = filt.x * (h.color.x * h.transp);
        filt.y = filt.y * (h.color.y * h.transp);
        filt.z = filt.z * (h.color.z * h.transp);
        /* dot(f,f) == 0 compiled as (z*z + y*y) == -(x*x) */
        if ((filt.z * filt.z + filt.y * filt.y) == -(filt.x * filt.x)) return V(0.0f, 0.0f, 0.0f);
        CNT(EV_HOP_CONTINUE);
        travelled = reach;
        sr.o = vadd(h.pos, step);
    }
    return filt;
}

/* The light-dependent part of phong_lighting (fragment_shader.glsl:491-513) for one light
 * position: (diffuse + specular) * attenuation.  Returns 0 when the light is fully blocked
 * (the reference then returns the ambient term alone, :493-495). */
static int phong_light_term(const scene_ctx *sc, v3 light, const hit_t *h, counters *cnt, v3 *term)
{
    v3 att = light_attenuation(sc, light, h->pos, cnt);
    if ((att.z * att.z + att.y * att.y) == -(att.x * att.x)) { CNT(EV_PHONG_SHADOWED); return 0; }
    CNT(EV_PHONG_FULL);
    v3 to_l = vsub(light, h->pos);
    v3 L = vscale(to_l, rsq(dot3(to_l, to_l)));
    float nl_z = h->n.z * L.z, nl_y = h->n.y * L.y, nl_x = h->n.x * L.x;
    float ndl = (nl_z + nl_y) + nl_x;
    float diff = fmaxf(ndl, 0.0f);
    v3 diffuse = vscale(h->color, diff);
    v3 to_v = vsub(sc->cam_pos, h->pos);
    v3 Vd = vscale(to_v, rsq(dot3(to_v, to_v)));
    /* reflect(-L, n) = -L - (2*dot(n,-L))*n with dot(n,-L) = ((-z) + (-y)) + (-x) */
    float dn = ((-nl_z) + (-nl_y)) + (-nl_x);
    float two_dn = dn * 2.0f;
    v3 R = V(-L.x - two_dn * h->n.x, -L.y - two_dn * h->n.y, -L.z - two_dn * h->n.z);
    float sd = fmaxf(dot3(Vd, R), 0.0f);
    float spec = 0.5f * orc_pow(sd, 32.0f);
    v3 sum = V(diffuse.x + spec, diffuse.y + spec, diffuse.z + spec);
    *term = V(sum.x * att.x, sum.y * att.y, sum.z * att.z);
    return 1;
}

/* phong_lighting, fragment_shader.glsl:488-514.  With light_grid G > 1 (not a reference
 * feature) the point light becomes a horizontal square of edge light_size centred on it, sampled
 * on a G x G grid: the light terms of the samples that are not fully blocked are added up in
 * sample order (column index fastest) starting from 0, and the mean joins the ambient term:
 * ambient + sum * (1 / G^2).  Ray paths, sky and ambient do not depend on the light, so this
 * is, up to rounding, the mean of G^2 point-light frames before the clamp. */
static v3 phong(const scene_ctx *sc, const hit_t *h, counters *cnt)
{
    v3 ambient = vscale(h->color, 0.05f);
    v3 term;
    if (sc->light_grid <= 1) {
        if (!phong_light_term(sc, sc->light, h, cnt, &term)) return ambient;
        return V(ambient.x + term.x, ambient.y + term.y, ambient.z + term.z);
    }
    v3 acc = V(0.0f, 0.0f, 0.0f);
    for (int j = 0; j < sc->light_grid; ++j)
        for (int i = 0; i < sc->light_grid; ++i) {
            v3 light = V(sc->light.x + sc->light_off[i], sc->light.y, sc->light.z + sc->light_off[j]);
            if (phong_light_term(sc, light, h, cnt, &term)) acc = V(acc.x + term.x, acc.y + term.y, acc.z + term.z);
        }
    return V(ambient.x + acc.x * sc->inv_light, ambient.y + acc.y * sc->inv_light, ambient.z + acc.z * sc->inv_light);
}

/* RayTraceIterative, fragment_shader.glsl:544-646 (with calculate_fresnel :524-534 inline).
 * capture_seg >= 0: record (id, t, depth) of that segment in visit order into cap[0..2]. */
static v3 ray_trace_iterative(const scene_ctx *sc, ray_t primary, counters *cnt,
                              int capture_seg, float *cap, int *nseg_out)
{
    v3 final = V(0.0f, 0.0f, 0.0f);
    raystate_t stack[ORC_MAX_STACK];
    int sp = 0, seg = 0;
    const int S = sc->stack_size;
    stack[sp].ray = primary; stack[sp].thr = V(1.0f, 1.0f, 1.0f);
    stack[sp].depth = 0; stack[sp].ior = 1.0f; sp++;
    while (sp > 0) {
        raystate_t cur = stack[--sp];
        hit_t hit = trace(sc, cur.ray, cnt);
        if (seg == capture_seg) { cap[0] = (float)hit.id; cap[1] = hit.hit ? hit.t : 0.0f; cap[2] = (float)cur.depth; }
        seg++;
        v3 d = cur.ray.d;
        if (!hit.hit) {                         /* sky, :570-580; mix(h,z,t) = h + t*(z-h) */
            CNT(EV_SEG_MISS);
            float t = 0.5f + 0.5f * d.y;
            float sx = 0.8f + t * (0.5f - 0.8f);
            float sy = 0.9f + t * (0.7f - 0.9f);
            final.x = final.x + sx * cur.thr.x;
            final.y = final.y + sy * cur.thr.y;
            final.z = final.z + cur.thr.z;       /* sky.z is exactly 1 */
            continue;
        }
        CNT(EV_SEG_HIT);
        /* dot(d,n) < 0 compiled as (z + y) < -(x) */
        int entering = (d.z * hit.n.z + d.y * hit.n.y) < -(d.x * hit.n.x);
        v3 on = entering ? hit.n : vneg(hit.n);
        float n1 = entering ? cur.ior : hit.ior;
        float n2 = entering ? hit.ior : 1.0f;
        float dni = dot3(d, on);
        float cos_t = fabsf(dni);
        /* calculate_fresnel */
        float r0 = (n1 - n2) / (n1 + n2); r0 = r0 * r0;
        float eta = n1 / n2;
        float sin_t2 = (eta * eta) * (1.0f - cos_t * cos_t);
        float F;
        if (sin_t2 > 1.0f) { CNT(EV_FRESNEL_TIR); F = 1.0f; }
        else { CNT(EV_FRESNEL_FULL); F = r0 + (1.0f - r0) * orc_pow(1.0f - cos_t, 5.0f); }
        float one_m_refl = 1.0f - hit.refl;
        float local = one_m_refl - hit.transp;
        if (local > 0.0f) {
            CNT(EV_LOCAL);
            v3 lc = phong(sc, &hit, cnt);
            final.x = final.x + (lc.x * local) * cur.thr.x;
            final.y = final.y + (lc.y * local) * cur.thr.y;
            final.z = final.z + (lc.z * local) * cur.thr.z;
        }
        if (cur.depth >= sc->max_depth - 1) continue;
        if (hit.transp > 0.0f) {                /* refraction, :612-626 */
            CNT(EV_REFR_TRY);
            float k = 1.0f - eta * (eta * (1.0f - dni * dni));
            v3 rdir = V(0.0f, 0.0f, 0.0f);
            if (!(k < 0.0f)) {
                float g = eta * dni + sqrtf(k);
                rdir = V(eta * d.x - g * on.x, eta * d.y - g * on.y, eta * d.z - g * on.z);
            }
            /* dot(r,r) > 0 compiled as -(z*z + y*y) < x*x */
            int valid = -(rdir.z * rdir.z + rdir.y * rdir.y) < rdir.x * rdir.x;
            if (valid && sp < S) {
                CNT(EV_REFR_PUSH);
                raystate_t ns;
                ns.ray.o = vadd(hit.pos, vscale(rdir, 0.001f));
                ns.ray.d = rdir;
                float omf = 1.0f - F;
                ns.thr = V(((cur.thr.x * omf) * hit.transp) * hit.color.x,
                           ((cur.thr.y * omf) * hit.transp) * hit.color.y,
                           ((cur.thr.z * omf) * hit.transp) * hit.color.z);
                ns.depth = cur.depth + 1; ns.ior = n2;
                stack[sp++] = ns;
            } else if (valid) CNT(EV_DROP);
        }
        CNT(EV_REFL_EVAL);
        float Rt = hit.refl + one_m_refl * F;   /* :629 */
        if (Rt > 0.0f) {
            if (sp < S) {
                CNT(EV_REFL_PUSH);
                raystate_t ns;
                ns.ray.o = vadd(hit.pos, vscale(on, 0.001f));
                float two_dni = dni * 2.0f;
                ns.ray.d = V(d.x - two_dni * on.x, d.y - two_dni * on.y, d.z - two_dni * on.z);
                ns.thr = vscale(cur.thr, Rt);
                ns.depth = cur.depth + 1; ns.ior = cur.ior;
                stack[sp++] = ns;
            } else CNT(EV_DROP);
        }
    }
    if (nseg_out) *nseg_out = seg;
    return final;
}

/* main(), fragment_shader.glsl:650-675, for the sample at window position (fx, fy). */
static ray_t primary_ray(const orc_params *p, float fx, float fy)
{
    const float *m = p->cam_to_world;
    v3 right = V(m[0], m[4], m[8]), up = V(m[1], m[5], m[9]), fwd = V(m[2], m[6], m[10]);
    float rw = (float)p->w, rh = (float)p->h;
    float uvx = (fx * 2.0f - rw) / rh;
    float uvy = (fy * 2.0f - rh) / rh;
    v3 dir = V((uvx * right.x + uvy * up.x) + p->focal * fwd.x,
               (uvx * right.y + uvy * up.y) + p->focal * fwd.y,
               (uvx * right.z + uvy * up.z) + p->focal * fwd.z);
    ray_t r; r.o = V(m[3], m[7], m[11]); r.d = normalize3(dir);
    return r;
}

/* Offset of sample i on an N-grid from the centre, in units of the grid's extent (sub-pixel
 * position of a stratified sample; position on the area light). */
static inline float sample_offset(int i, int N) { return (float)(((double)i + 0.5) / (double)N - 0.5); }

static int check_args(const orc_object *objs, int n, const orc_params *p)
{
    if (!objs || !p || n < 0) return -1;
    if (p->w <= 0 || p->h <= 0) return -2;
    if (p->stack_size < 1 || p->stack_size > ORC_MAX_STACK) return -3;
    if (p->max_depth < 1 || p->shadow_hops < 0 || p->spp_grid < 1) return -4;
    if (p->light_grid < 0 || p->light_grid > ORC_MAX_LIGHT_GRID || !(p->light_size >= 0.0f)) return -6;
    for (int i = 0; i < n; ++i) if (objs[i].type != 1 && objs[i].type != 2) return -5;
    return 0;
}

static void make_ctx(scene_ctx *sc, const orc_object *objs, int n, const orc_params *p)
{
    sc->objs = objs; sc->n = n;
    sc->light = V(p->light[0], p->light[1], p->light[2]);
    sc->cam_pos = V(p->cam_to_world[3], p->cam_to_world[7], p->cam_to_world[11]);
    sc->max_depth = p->max_depth; sc->stack_size = p->stack_size; sc->hops = p->shadow_hops;
    sc->light_grid = p->light_grid > 1 ? p->light_grid : 1;
    for (int i = 0; i < ORC_MAX_LIGHT_GRID; ++i)
        sc->light_off[i] = i < sc->light_grid ? sample_offset(i, sc->light_grid) * p->light_size : 0.0f;
    sc->inv_light = 1.0f / (float)(sc->light_grid * sc->light_grid);
    sc->polish = (p->flags & ORC_FLAG_NEWTON_POLISH) ? 1 : 0;
}


/* Framebuffer write: clamp to [0,1], scale by 255, round to nearest even (RGBA8 UNORM). */
static inline uint8_t to_unorm8(float c)
{
    if (!(c > 0.0f)) return 0;
    if (c >= 1.0f) return 255;
    return (uint8_t)(int)nearbyintf(c * 255.0f);
}

/*
 * Render rows [y0, y1) of the frame (y in GL convention: 0 = bottom row).
 *   out_f32: optional, (y1-y0)*w*4 floats, unclamped colour, alpha 1
 *   out_u8 : optional, (y1-y0)*w*4 bytes, RGBA8 as the framebuffer would hold it
 *   ev_out : optional, EV_COUNT uint64 event counters (summed over the rendered rows)
 * Returns 0 or a negative error code.
 */
int orc_render_rows(const orc_object *objs, int n, const orc_params *p, int y0, int y1,
                    float *out_f32, uint8_t *out_u8, uint64_t *ev_out, int nthreads)
{
    int rc = check_args(objs, n, p);
    if (rc) return rc;
    if (y0 < 0 || y1 > p->h || y0 > y1) return -6;
    scene_ctx sc; make_ctx(&sc, objs, n, p);
    const int N = p->spp_grid, w = p->w;
    const float inv_spp = 1.0f / (float)(N * N);
    counters total; memset(&total, 0, sizeof total);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        _mm_setcsr(_mm_getcsr() | 0x8040);     /* FTZ + DAZ */
        counters local; memset(&local, 0, sizeof local);
        counters *cnt = ev_out ? &local : NULL;
#pragma omp for schedule(dynamic, 4)
        for (int y = y0; y < y1; ++y) {
            for (int x = 0; x < w; ++x) {
                v3 acc = V(0.0f, 0.0f, 0.0f);
                for (int sj = 0; sj < N; ++sj)
                    for (int si = 0; si < N; ++si) {
                        float fx = ((float)x + 0.5f) + sample_offset(si, N);
                        float fy = ((float)y + 0.5f) + sample_offset(sj, N);
                        if (cnt) cnt->c[EV_PIXEL_PATH]++;
                        v3 c = ray_trace_iterative(&sc, primary_ray(p, fx, fy), cnt, -1, NULL, NULL);
                        if (N == 1) acc = c; else acc = vadd(acc, c);
                    }
                if (N > 1) acc = vscale(acc, inv_spp);
                size_t o = ((size_t)(y - y0) * w + x) * 4;
                if (out_f32) { out_f32[o] = acc.x; out_f32[o + 1] = acc.y; out_f32[o + 2] = acc.z; out_f32[o + 3] = 1.0f; }
                if (out_u8) { out_u8[o] = to_unorm8(acc.x); out_u8[o + 1] = to_unorm8(acc.y); out_u8[o + 2] = to_unorm8(acc.z); out_u8[o + 3] = 255; }
            }
        }
        if (ev_out) {
#pragma omp critical
            for (int i = 0; i < EV_COUNT; ++i) total.c[i] += local.c[i];
        }
    }
    if (ev_out) memcpy(ev_out, total.c, sizeof total.c);
    return 0;
}

/* Whole frame. */
int orc_render(const orc_object *objs, int n, const orc_params *p,
               float *out_f32, uint8_t *out_u8, uint64_t *ev_out, int nthreads)
{
    if (!p) return -1;
    return orc_render_rows(objs, n, p, 0, p->h, out_f32, out_u8, ev_out, nthreads);
}

/* For segment `seg` (visit order, 0 = primary) of every pixel-centre path write
 * (id, t, depth, number of segments); id -1 sky, 0..n-1 object, n plane, -2 no such segment. */
int orc_capture(const orc_object *objs, int n, const orc_params *p, int seg, float *out4, int nthreads)
{
    int rc = check_args(objs, n, p);
    if (rc) return rc;
    scene_ctx sc; make_ctx(&sc, objs, n, p);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        _mm_setcsr(_mm_getcsr() | 0x8040);
#pragma omp for schedule(dynamic, 4)
        for (int y = 0; y < p->h; ++y)
            for (int x = 0; x < p->w; ++x) {
                float cap[3] = { -2.0f, 0.0f, 0.0f }; int nseg = 0;
                ray_trace_iterative(&sc, primary_ray(p, (float)x + 0.5f, (float)y + 0.5f), NULL, seg, cap, &nseg);
                float *o = out4 + ((size_t)y * p->w + x) * 4;
                o[0] = cap[0]; o[1] = cap[1]; o[2] = cap[2]; o[3] = (float)nseg;
            }
    }
    return 0;
}

int orc_event_count(void) { return EV_COUNT; }
int orc_event_flops(int ev) { return (ev >= 0 && ev < EV_COUNT) ? ev_flops[ev] : -1; }
