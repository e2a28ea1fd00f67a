// rt_numerics.h -- arithmetic primitives of the render kernel, per numerics flavour.
// Included by rt_device.hip after RT_NS and DEV are defined; not a stand-alone header.
#pragma once

namespace RT_NS {

// ------------------------------------------------------------------ numerics

DEV float bits_f(uint32_t u) { return __uint_as_float(u); }
DEV uint32_t f_bits(float f) { return __float_as_uint(f); }

#ifdef RT_FAST
// fast flavour: hardware reciprocal / rsqrt / sqrt (1 ulp), FMA contraction (compile flag)
DEV float rt_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
DEV float rt_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
DEV float rt_rsq(float a) { return __builtin_amdgcn_rsqf(a); }
template <int C> DEV float rt_div_const(float x) { return x * (1.0f / (float)C); }
DEV float div_by_reciprocal(float x, float d, float rd) { return x * rd; }
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
// x / d given rd = RN(1/d): the same three instructions with a divisor known at run time.  Whether
// the result is the correctly rounded quotient depends on (x, d); callers use it only for operand
// sets the host has checked one by one with this very sequence (uv_quotients_by_reciprocal in
// rt_api.hip).
DEV float div_by_reciprocal(float x, float d, float rd)
{
    float q = x * rd;
    float r = __builtin_fmaf(-d, q, x);
    return __builtin_fmaf(r, rd, q);
}
DEV bool div_const_in_core_range(float x) { return ((f_bits(x) & 0x7fffffffu) - 0x0f800000u) < 0x60000000u; }
template <int C> DEV float rt_div_const(float x)
{
    if (div_const_in_core_range(x)) return div_const_core<C>(x);
    return x / (float)C;        // [census-cold]
}
// binary32 patterns 0x0f800000 .. 0x6f7fffff are the positive numbers in [2^-96, 2^96)
DEV bool sqrt_in_core_range(float x) { return (f_bits(x) - 0x0f800000u) < 0x60000000u; }
DEV float rt_sqrt(float a)
{
    if (sqrt_in_core_range(a)) return sqrt_core(a);
    return __builtin_sqrtf(a);      // [census-cold]
}
DEV float rt_rsq(float a)
{
    if (sqrt_in_core_range(a)) return rcp_core(sqrt_core(a));
    return 1.0f / __builtin_sqrtf(a);      // [census-cold]
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

}  // namespace RT_NS
