/* pc_math.h -- the float32 functions of the bitstream contract.
 *
 * In this codec the entropy-coded symbols are round(y - mu) of float32 network
 * outputs, and the decoder must re-derive mu / scale / mask bit-for-bit from what
 * it has decoded (reference: CHProg_cnn.py:740-762 vs :880-901).  Every float
 * function on that path is therefore part of the format: encoder and decoder --
 * on whatever device -- must evaluate it identically.  This header defines those
 * functions with IEEE-754 basic operations only (+ - * / sqrt fma floor and bit
 * casts, all correctly rounded on x86-64 and on gfx950), so the HIP kernels
 * (hipcc) and the CPU oracle (gcc) produce identical bits.
 *
 * Build rules (enforced by the Makefiles): -ffp-contract=off on both compilers
 * (every fused multiply-add below is an explicit fmaf), no -ffast-math, HIP's
 * default correctly-rounded f32 divide/sqrt, f32 denormals enabled (gfx950 default).
 *
 * Accuracy against a float64 reference (tests/test_pc_math.py): expf <= 1 ulp,
 * erff <= 2 ulp, tanhf <= 2 ulp.  They replace, on this path only, the libm / SLEEF
 * functions behind torch.erf (nn.GELU, layers.py:45), torch.tanh (CHProg_cnn.py:761),
 * torch.sigmoid (layers.py:73) and softmax's exp (win_attention.py:82).
 */
#ifndef PC_MATH_H
#define PC_MATH_H

/* Identifies the numeric contract a bitstream was coded under: high half = revision of the contraction order
 * (2: aligned groups of 8 k visited 0,4,1,5,2,6,3,7, one fmaf chain per output element, bias added afterwards), low half = revision of
 * the functions below.  A decoder built to another contract re-derives other mu / scale bits and desynchronises: containers carry
 * this id and are refused on mismatch (progressivecodec_amd/container.py); pc_contract_id() exports it. */
#define PC_NUMERIC_CONTRACT_ID 0x00020001u

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PC_HD __host__ __device__ static inline
#else
#define PC_HD static inline
#endif

PC_HD float pc_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
PC_HD uint32_t pc_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp(x), |rel err| <= 1 ulp on [-87, 88]; exactly 0 below -87, +inf above 88.7. */
PC_HD float pc_expf(float x)
{
    if (!(x > -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.7f) return pc_bits2f(0x7f800000u);
    const float n = floorf(fmaf(x, 1.442695040888963387f, 0.5f));
    float r = fmaf(n, -6.931152344e-01f, x);
    r = fmaf(n, -3.194618495e-05f, r);
    float p = 1.989939192e-04f;
    p = fmaf(p, r, 1.393373357e-03f);
    p = fmaf(p, r, 8.333298378e-03f);
    p = fmaf(p, r, 4.166646302e-02f);
    p = fmaf(p, r, 1.666666716e-01f);
    p = fmaf(p, r, 5.000000000e-01f);
    const float r2 = r * r;
    const float e = fmaf(p, r2, r) + 1.0f;       /* exp(r), r in [-ln2/2, ln2/2] */
    const int ni = (int)n;                        /* -125 .. 128 */
    if (ni > 127) return (e * 2.0f) * pc_bits2f((uint32_t)(ni - 1 + 127) << 23);
    return e * pc_bits2f((uint32_t)(ni + 127) << 23);
}

/* erf(x), <= 2 ulp. */
PC_HD float pc_erff(float x)
{
    const float a = fabsf(x);
    if (a < 1.0f) {
        const float t = x * x;
        float r = 1.059527904e-06f;
        r = fmaf(r, t, -1.391170190e-05f);
        r = fmaf(r, t, 1.195694713e-04f);
        r = fmaf(r, t, -8.542670403e-04f);
        r = fmaf(r, t, 5.223785061e-03f);
        r = fmaf(r, t, -2.686613426e-02f);
        r = fmaf(r, t, 1.128379107e-01f);
        r = fmaf(r, t, -3.761263788e-01f);
        r = fmaf(r, t, 1.283791661e-01f);
        return fmaf(x, r, x);
    }
    if (!(a < 4.0f)) return (x != x) ? x : (x > 0.0f ? 1.0f : -1.0f);
    const float u = a - 2.5f;
    float q = 1.777464753e-08f;
    q = fmaf(q, u, -4.104854057e-08f);
    q = fmaf(q, u, -1.758092054e-07f);
    q = fmaf(q, u, 1.844959684e-06f);
    q = fmaf(q, u, -1.272867667e-05f);
    q = fmaf(q, u, 7.693984662e-05f);
    q = fmaf(q, u, -4.209505278e-04f);
    q = fmaf(q, u, 2.165525686e-03f);
    q = fmaf(q, u, -1.085806731e-02f);
    q = fmaf(q, u, 5.610625818e-02f);
    q = fmaf(q, u, -3.526807427e-01f);
    q = fmaf(q, u, -1.556815267e+00f);
    const float erfc_a = pc_expf(fmaf(-a, a, q));  /* erfc(a) = exp(-a^2 + q(a)) */
    const float r = 1.0f - erfc_a;
    return x > 0.0f ? r : -r;
}

/* tanh(x), <= 2 ulp. */
PC_HD float pc_tanhf(float x)
{
    const float a = fabsf(x);
    if (a < 0.625f) {
        const float t = x * x;
        float p = -8.647738723e-04f;
        p = fmaf(p, t, 3.309384221e-03f);
        p = fmaf(p, t, -8.792471141e-03f);
        p = fmaf(p, t, 2.186022140e-02f);
        p = fmaf(p, t, -5.396767333e-02f);
        p = fmaf(p, t, 1.333333254e-01f);
        p = fmaf(p, t, -3.333333433e-01f);
        return fmaf(x * t, p, x);
    }
    if (!(a < 9.5f)) return (x != x) ? x : (x > 0.0f ? 1.0f : -1.0f);
    const float e = pc_expf(2.0f * a);
    const float r = 1.0f - 2.0f / (e + 1.0f);
    return x > 0.0f ? r : -r;
}

/* logistic sigmoid, 1 / (1 + exp(-x)) -- the formula ATen uses on CPU. */
PC_HD float pc_sigmoidf(float x) { return 1.0f / (1.0f + pc_expf(-x)); }

/* exact (erf) GELU: nn.GELU() default, 0.5 x (1 + erf(x / sqrt 2)). */
PC_HD float pc_geluf(float x) { return (0.5f * x) * (1.0f + pc_erff(x * 0.70710678118654752440f)); }

/* torch.rsqrt on CPU is 1 / sqrt(x) (two correctly rounded operations). */
PC_HD float pc_rsqrtf(float x) { return 1.0f / sqrtf(x); }

/* round half to even (torch.round), exact for |x| < 2^23, identity above. */
PC_HD float pc_roundevenf(float x)
{
    const float a = fabsf(x);
    if (!(a < 8388608.0f)) return x;
    const float r = (a + 8388608.0f) - 8388608.0f;  /* RN-even add trick; no contraction possible */
    return x < 0.0f ? -r : (x == 0.0f ? x : r);
}

#endif /* PC_MATH_H */
