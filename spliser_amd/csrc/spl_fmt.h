// spl_fmt.h -- numbers as text, the way the reference's Python writes them (outputBedFile SpliSER_v0_1_8.py:641-664,
// outputCombinedLines :722-740): "%d", "{0:.3f}" / "{0:.5f}" and str(float).  Host only; shared by the .SpliSER.tsv writer
// (spl_host.cpp) and the .combined.tsv writer (spl_combine.cpp).
#ifndef SPL_FMT_H
#define SPL_FMT_H
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace splfmt {

// "%d" of v into out; returns the length.
inline size_t fmt_int(char *out, int64_t v)
{
    char tmp[24];
    int n = 0;
    uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    do { tmp[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
    size_t k = 0;
    if (v < 0) out[k++] = '-';
    while (n) out[k++] = tmp[--n];
    return k;
}

// "%.<digits>f" of x (digits <= 6) into out, exactly as printf and Python's format() round: to the nearest decimal of the
// double's EXACT binary value, ties to even.  x = m * 2^e with an integer m: m * 10^digits is split at the binary point with
// integer arithmetic, nothing is ever rounded before the one decision that matters.  Values this does not cover (negative,
// not finite, 2^63 / 10^digits and beyond) go to snprintf.
inline size_t fmt_fixed(char *out, double x, int digits)
{
    static const uint64_t pow10[7] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull};
    if (!(x >= 0.0) || !(x < 9.0e12) || digits > 6) return (size_t)snprintf(out, 64, "%.*f", digits, x);
    int e = 0;
    const double f = frexp(x, &e);                    // x = f * 2^e, f in [0.5, 1) (or 0)
    const uint64_t m = (uint64_t)ldexp(f, 53);         // 53-bit integer, exact
    const int sh = 53 - e;                             // x = m / 2^sh
    unsigned __int128 v = (unsigned __int128)m * pow10[digits];
    uint64_t q;
    if (sh <= 0) {
        q = (uint64_t)(v << (-sh));                    // an integer already (x < 9e12: fits)
    } else if (sh >= 120) {
        q = 0;                                         // far below half a unit in the last place
    } else {
        const unsigned __int128 one = (unsigned __int128)1 << sh;
        const unsigned __int128 rem = v & (one - 1), half = one >> 1;
        q = (uint64_t)(v >> sh);
        if (rem > half || (rem == half && (q & 1u))) ++q;
    }
    const uint64_t ip = q / pow10[digits], fp = q % pow10[digits];
    size_t k = fmt_int(out, (int64_t)ip);
    if (digits) {
        out[k++] = '.';
        for (int d = digits - 1; d >= 0; --d) out[k++] = (char)('0' + (fp / pow10[d]) % 10u);
    }
    return k;
}


// str(x) of a Python float (repr: the shortest digits that read back as x; fixed notation for 1e-4 <= |x| < 1e16, ".0" behind
// an integer, otherwise d.ddde+XX with at least two exponent digits) into out (32 bytes are enough); returns the length.
inline size_t fmt_repr(char *out, double x)
{
    if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) { const size_t n = x < 0 ? 4 : 3; memcpy(out, x < 0 ? "-inf" : "inf", n); return n; }
    size_t k = 0;
    if (std::signbit(x)) { out[k++] = '-'; x = -x; }
    if (x == 0.0) { memcpy(out + k, "0.0", 3); return k + 3; }
    char sci[40]; // shortest round-trip digits, "d[.ddd]e[+-]XX"
    const std::to_chars_result r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);
    const char *e = sci;
    while (e < r.ptr && *e != 'e') ++e;
    char dig[24];
    int nd = 0;
    for (const char *q = sci; q < e; ++q) if (*q != '.') dig[nd++] = *q;
    int exp10 = 0;
    {
        const char *q = e + 1;
        const bool neg = *q == '-';
        if (*q == '+' || *q == '-') ++q;
        for (; q < r.ptr; ++q) exp10 = exp10 * 10 + (*q - '0');
        if (neg) exp10 = -exp10;
    }
    if (exp10 >= -4 && exp10 < 16) {
        if (exp10 < 0) {
            out[k++] = '0'; out[k++] = '.';
            for (int z = -1; z > exp10; --z) out[k++] = '0';
            for (int d = 0; d < nd; ++d) out[k++] = dig[d];
        } else {
            for (int d = 0; d <= exp10; ++d) out[k++] = d < nd ? dig[d] : '0';
            out[k++] = '.';
            if (nd > exp10 + 1) for (int d = exp10 + 1; d < nd; ++d) out[k++] = dig[d];
            else out[k++] = '0';
        }
        return k;
    }
    out[k++] = dig[0];
    if (nd > 1) { out[k++] = '.'; for (int d = 1; d < nd; ++d) out[k++] = dig[d]; }
    out[k++] = 'e';
    out[k++] = exp10 < 0 ? '-' : '+';
    const int a = exp10 < 0 ? -exp10 : exp10;
    if (a >= 100) out[k++] = (char)('0' + a / 100);
    out[k++] = (char)('0' + a / 10 % 10);
    out[k++] = (char)('0' + a % 10);
    return k;
}

} // namespace splfmt
#endif
