/* Exhaustive check that the f64-free form of float_sample_to_i16 used on the device (csrc/sk_device.h,
 * dev_float_sample_to_i16_f32, and its shorter sibling dev_float_sample_to_i16_v2) equals the reference's f64 form (soundkit-decoder lib.rs:1815-1827) bit for bit.
 *   gcc -O2 -ffp-contract=off -o check tools/check_f32_rounding.c -lm
 *   ./check FIRST_PART N_PARTS      (the 2^32 bit patterns are split into 16 parts of 2^28; 0 16 = all of them)
 * All 16 parts: 0 mismatches (run when the function was written); tests/test_oracle_pins.py runs the binades that
 * contain every rounding boundary (|x| in [2^-20, 2]) on every CPU run. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int reference_form(float x) {
    float f = isfinite(x) ? fminf(fmaxf(x, -1.0f), 1.0f) : 0.0f;
    double scaled = f < 0.0f ? (double)f * 32768.0 : (double)f * 32767.0;
    int r = (int)round(scaled);
    return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

static int f32_form(float x) { /* keep in step with dev_float_sample_to_i16_f32 */
    float m = isfinite(x) ? fminf(fabsf(x), 1.0f) : 0.0f;
    float a = m * 32768.0f;
    float k = floorf(a);
    float d = a - k;
    int ki = (int)k;
    int neg = ki + (d >= 0.5f ? 1 : 0);
    int pos = ki + ((d - 0.5f >= m) ? 1 : 0) - ((d + 0.5f < m) ? 1 : 0);
    return x < 0.0f ? -neg : pos;
}

static int f32_form_v2(float x) { /* keep in step with dev_float_sample_to_i16_v2 (the synthesis kernel's s16 output) */
    float m = isfinite(x) ? fminf(fabsf(x), 1.0f) : 0.0f;
    float a = m * 32768.0f;
    float d = a - floorf(a); /* v_fract_f32: exact here (a < 2^15 + 1) */
    int ki = (int)a;
    float mm = x < 0.0f ? 0.0f : m;
    int r = ki + ((d - 0.5f >= mm) ? 1 : 0) - ((d + 0.5f < mm) ? 1 : 0);
    return x < 0.0f ? -r : r;
}

/* v_cvt_i32_f64: truncates, saturates, NaN -> 0;  v_cvt_pk_i16_i32: saturates to i16 */
static int cvt_i32_f64(double t) {
    if (t != t) return 0;
    if (t >= 2147483647.0) return 2147483647;
    if (t <= -2147483648.0) return (int)-2147483648LL;
    return (int)t;
}
static int sat16(int k) { return k < -32768 ? -32768 : (k > 32767 ? 32767 : k); }

static int f64_form_v3(float x) { /* keep in step with dev_float_sample_to_i16_sat (sk_device.h) */
    float y = fmaf(x, 0.0f, x); /* NaN for +-inf and NaN, x otherwise */
    double X = (double)y;
    uint32_t b;
    memcpy(&b, &x, 4);
    double half = (b >> 31) ? -0.5 : 0.5;
    double h = fma(fabs(X), -0.5, half);
    double t = fma(X, 32767.5, h); /* = x * 32767 (x >= 0) or x * 32768 (x < 0), plus half away from zero: exact for |x| <= 2 */
    return sat16(cvt_i32_f64(t));
}

static uint64_t sweep(uint64_t lo, uint64_t hi) {
    uint64_t bad = 0;
    for (uint64_t u = lo; u < hi; ++u) {
        uint32_t b = (uint32_t)u;
        float x;
        memcpy(&x, &b, 4);
        if (reference_form(x) != f32_form(x) || reference_form(x) != f32_form_v2(x) || reference_form(x) != f64_form_v3(x)) {
            if (bad < 10) printf("mismatch %a: %d vs %d / %d / %d\n", x, reference_form(x), f32_form(x), f32_form_v2(x), f64_form_v3(x));
            ++bad;
        }
    }
    return bad;
}

int main(int argc, char **argv) {
    uint64_t bad = 0;
    if (argc > 1 && strcmp(argv[1], "boundaries") == 0) {
        /* exponents 2^-20 .. 2^1, both signs, plus zeros / denormals' first binade / inf / nan patterns */
        for (uint32_t sign = 0; sign < 2; ++sign) {
            for (uint32_t e = 127 - 20; e <= 127 + 1; ++e)
                bad += sweep(((uint64_t)sign << 31) | ((uint64_t)e << 23), (((uint64_t)sign << 31) | ((uint64_t)(e + 1) << 23)));
            bad += sweep(((uint64_t)sign << 31), ((uint64_t)sign << 31) + (1u << 16));
            bad += sweep(((uint64_t)sign << 31) | 0x7f800000u, (((uint64_t)sign << 31) | 0x7f800000u) + (1u << 16));
        }
    } else {
        uint32_t part = argc > 1 ? (uint32_t)atoi(argv[1]) : 0, parts = argc > 2 ? (uint32_t)atoi(argv[2]) : 16;
        bad = sweep((uint64_t)part << 28, (uint64_t)(part + parts) << 28);
    }
    printf("%llu mismatches\n", (unsigned long long)bad);
    return bad != 0;
}
