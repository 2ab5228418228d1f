// json_text.hpp — number and string TEXT of the OpenMVG file the way the reference's JSON library produces it.
//
// The reference writes its output with rapidjson's PrettyWriter (output_sfm_data.cpp:186-229) after reading the
// input with rapidjson's reader (default flags, OpenMvgParser.cpp:49-50, output_sfm_data.cpp:187-193): a float
// coordinate is widened to double and printed by Grisu2 ("Printing Floating-Point Numbers Quickly and Accurately
// with Integers", Loitsch, PLDI 2010) followed by a fixed notation rule; a number COPIED from the input file is
// re-printed from what the reader made of it (integer kinds stay integers; everything else is converted to the
// NEAREST double — the reference's vendored copy of the library sets kParseFullPrecisionFlag as its default,
// external/rapidjson/reader.h:137 — then printed by Grisu2). For the written file to be byte-identical to the
// reference's, both are restated here from the published algorithm and the library's documented behaviour; the
// cached powers of ten of Grisu are COMPUTED (exact integer arithmetic, round to nearest) instead of tabulated.
// tests/test_json_rapidjson.py checks text and table against the vendored rapidjson of the reference tree when
// that tree is present.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <locale.h>
#include <string>
#include <vector>

namespace eg3d_json {

// ---------------------------------------------------------------- 64-bit "do it yourself" floats
struct Fp {
  uint64_t f = 0;
  int e = 0;
};
inline Fp fp_mul(const Fp& a, const Fp& b) {  // upper 64 bits of the 128-bit product, rounded half up
  const unsigned __int128 p = (unsigned __int128)a.f * b.f;
  uint64_t h = (uint64_t)(p >> 64);
  if ((uint64_t)p & (1ull << 63)) h++;
  Fp r;
  r.f = h;
  r.e = a.e + b.e + 64;
  return r;
}
inline Fp fp_normalize(Fp v) {
  while (!(v.f & (1ull << 63))) {
    v.f <<= 1;
    v.e--;
  }
  return v;
}
inline Fp fp_from_double(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  const int be = (int)((u >> 52) & 0x7ff);
  const uint64_t m = u & ((1ull << 52) - 1);
  Fp v;
  if (be) {
    v.f = m + (1ull << 52);
    v.e = be - 0x3ff - 52;
  } else {
    v.f = m;
    v.e = 1 - 0x3ff - 52;
  }
  return v;
}

// ---------------------------------------------------------------- cached powers 10^(-348 + 8 i), i = 0 .. 86
// exact big-integer arithmetic: little-endian 32-bit limbs
struct Big {
  std::vector<uint32_t> w;
  void mul_small(uint32_t m) {
    uint64_t c = 0;
    for (auto& x : w) {
      c += (uint64_t)x * m;
      x = (uint32_t)c;
      c >>= 32;
    }
    if (c) w.push_back((uint32_t)c);
  }
  int bits() const {
    for (int i = (int)w.size() - 1; i >= 0; i--)
      if (w[i]) return i * 32 + 32 - __builtin_clz(w[i]);
    return 0;
  }
  bool bit(int i) const { return i >= 0 && (size_t)(i >> 5) < w.size() && ((w[i >> 5] >> (i & 31)) & 1u); }
  bool any_below(int i) const {  // any set bit in positions [0, i)
    for (int k = 0; k < i; k++)
      if (bit(k)) return true;
    return false;
  }
  void shl1() {
    uint32_t c = 0;
    for (auto& x : w) {
      const uint32_t n = x >> 31;
      x = (x << 1) | c;
      c = n;
    }
    if (c) w.push_back(c);
  }
  int cmp(const Big& o) const {
    const size_t n = w.size() > o.w.size() ? w.size() : o.w.size();
    for (size_t i = n; i-- > 0;) {
      const uint32_t a = i < w.size() ? w[i] : 0, b = i < o.w.size() ? o.w[i] : 0;
      if (a != b) return a < b ? -1 : 1;
    }
    return 0;
  }
  void sub(const Big& o) {  // this >= o
    int64_t c = 0;
    for (size_t i = 0; i < w.size(); i++) {
      int64_t v = (int64_t)w[i] - (i < o.w.size() ? o.w[i] : 0) + c;
      c = v < 0 ? -1 : 0;
      w[i] = (uint32_t)(v + (c ? (1ll << 32) : 0));
    }
  }
};
inline Big big_pow10(int n) {
  Big b;
  b.w.push_back(1);
  for (int i = 0; i < n; i++) b.mul_small(10);
  return b;
}
inline Fp cached_power_compute(int k) {  // 10^k as a normalised 64-bit significand (nearest) and binary exponent
  Fp r;
  if (k >= 0) {
    const Big p = big_pow10(k);
    const int nb = p.bits();
    if (nb <= 64) {
      uint64_t f = 0;
      for (int i = nb - 1; i >= 0; i--) f = (f << 1) | (p.bit(i) ? 1u : 0u);
      r.f = f << (64 - nb);
      r.e = nb - 64;
      return r;
    }
    uint64_t f = 0;
    for (int i = nb - 1; i >= nb - 64; i--) f = (f << 1) | (p.bit(i) ? 1u : 0u);
    r.e = nb - 64;
    if (p.bit(nb - 65)) {  // round to nearest (an exact tie cannot occur: 10^k has few trailing zero bits)
      f++;
      if (f == 0) {
        f = 1ull << 63;
        r.e++;
      }
    }
    r.f = f;
    return r;
  }
  // 10^k = 1 / 10^m: binary long division of a power of two by D = 10^m, 64 quotient bits + a rounding bit
  const Big D = big_pow10(-k);
  const int nb = D.bits();
  Big R;  // remainder, starts as 2^(nb-1) (<= D), quotient bits from the first position that gives a 1
  R.w.assign((size_t)(nb + 31) / 32 + 1, 0);
  R.w[(size_t)(nb - 1) >> 5] = 1u << ((nb - 1) & 31);
  int shift = nb - 1;  // R = 2^shift at the start
  uint64_t q = 0;
  int produced = 0;
  bool started = false;
  bool round_bit = false;
  while (produced < 65) {
    bool one = false;
    if (R.cmp(D) >= 0) {
      R.sub(D);
      one = true;
    }
    if (one) started = true;
    if (started) {
      if (produced < 64)
        q = (q << 1) | (one ? 1u : 0u);
      else
        round_bit = one;
      produced++;
    }
    if (produced < 65) {
      R.shl1();
      shift++;
    }
  }
  // quotient bit j (0 = first) has weight 2^(first_shift - j) / ... : value = q * 2^(e) with
  // e = (nb - 1) - shift_at_first_one ... derive from the invariant 2^shift = Q * D + R  =>  after the loop the
  // 65 bits produced are floor(2^shift / D) truncated to its top 65 bits, i.e. 1/D ~= (q:round) * 2^(-shift) * 2
  // (the last produced bit has weight 2^0 of floor(2^shift / D)).
  r.e = -shift + 1;  // the 64-bit q ends one bit above the rounding bit
  if (round_bit) {
    q++;
    if (q == 0) {
      q = 1ull << 63;
      r.e++;
    }
  }
  r.f = q;
  return r;
}
inline const Fp* cached_powers() {
  static Fp table[87];
  static bool ready = false;
  if (!ready) {
    for (int i = 0; i < 87; i++) table[i] = cached_power_compute(-348 + 8 * i);
    ready = true;
  }
  return table;
}
inline Fp cached_power_for(int e, int* K) {  // the power that brings a binary exponent e into Grisu's window
  const double dk = (-61 - e) * 0.30102999566398114 + 347;  // 1 / log2(10)
  int k = (int)dk;
  if (dk - k > 0.0) k++;
  const unsigned index = (unsigned)((k >> 3) + 1);
  *K = -(-348 + (int)(index << 3));
  return cached_powers()[index];
}

// ---------------------------------------------------------------- Grisu2 digit generation
inline void grisu_round(char* buf, int len, uint64_t delta, uint64_t rest, uint64_t ten_kappa, uint64_t wp_w) {
  while (rest < wp_w && delta - rest >= ten_kappa &&
         (rest + ten_kappa < wp_w || wp_w - rest > rest + ten_kappa - wp_w)) {
    buf[len - 1]--;
    rest += ten_kappa;
  }
}
inline int count_digits_u32(uint32_t n) {
  if (n < 10) return 1;
  if (n < 100) return 2;
  if (n < 1000) return 3;
  if (n < 10000) return 4;
  if (n < 100000) return 5;
  if (n < 1000000) return 6;
  if (n < 10000000) return 7;
  if (n < 100000000) return 8;
  return 9;  // more cannot occur in digit generation
}
inline void digit_gen(const Fp& W, const Fp& Mp, uint64_t delta, char* buf, int* len, int* K) {
  static const uint32_t p10[] = {1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000, 1000000000};
  Fp one;
  one.f = 1ull << -Mp.e;
  one.e = Mp.e;
  const uint64_t wp_w = Mp.f - W.f;
  uint32_t p1 = (uint32_t)(Mp.f >> -one.e);
  uint64_t p2 = Mp.f & (one.f - 1);
  int kappa = count_digits_u32(p1);
  *len = 0;
  while (kappa > 0) {
    const uint32_t div = p10[kappa - 1];
    const uint32_t d = p1 / div;
    p1 %= div;
    if (d || *len) buf[(*len)++] = (char)('0' + d);
    kappa--;
    const uint64_t tmp = ((uint64_t)p1 << -one.e) + p2;
    if (tmp <= delta) {
      *K += kappa;
      grisu_round(buf, *len, delta, tmp, (uint64_t)p10[kappa] << -one.e, wp_w);
      return;
    }
  }
  for (;;) {  // kappa = 0: the fractional part
    p2 *= 10;
    delta *= 10;
    const char d = (char)(p2 >> -one.e);
    if (d || *len) buf[(*len)++] = (char)('0' + d);
    p2 &= one.f - 1;
    kappa--;
    if (p2 < delta) {
      *K += kappa;
      const int index = -kappa;
      // The weight of wp_w at this digit position is 10^index. The rapidjson the reference vendors indexes its
      // ten-entry table with `index` unchecked: index 9 takes 10^9 (later releases of the library use 0 from
      // index 9 on), and an index >= 10 reads past the table — undefined there; the build of that library made
      // for tests/test_json_rapidjson.py finds zeros, which is also what later releases define. Followed here.
      grisu_round(buf, *len, delta, p2, one.f, wp_w * (index <= 9 ? p10[index] : 0));
      return;
    }
  }
}
inline void grisu2(double value, char* buf, int* len, int* K) {  // value > 0
  const Fp v = fp_from_double(value);
  // boundaries m-, m+ of the rounding interval, on a common exponent
  Fp pl;
  pl.f = (v.f << 1) + 1;
  pl.e = v.e - 1;
  while (!(pl.f & (1ull << 53))) {
    pl.f <<= 1;
    pl.e--;
  }
  pl.f <<= 64 - 52 - 2;
  pl.e -= 64 - 52 - 2;
  Fp mi;
  if (v.f == (1ull << 52)) {
    mi.f = (v.f << 2) - 1;
    mi.e = v.e - 2;
  } else {
    mi.f = (v.f << 1) - 1;
    mi.e = v.e - 1;
  }
  mi.f <<= mi.e - pl.e;
  mi.e = pl.e;
  const Fp c = cached_power_for(pl.e, K);
  const Fp W = fp_mul(fp_normalize(v), c);
  Fp Wp = fp_mul(pl, c), Wm = fp_mul(mi, c);
  Wm.f++;
  Wp.f--;
  digit_gen(W, Wp, Wp.f - Wm.f, buf, len, K);
}

// ---------------------------------------------------------------- notation (what follows the digits)
inline char* write_exponent(int K, char* p) {
  if (K < 0) {
    *p++ = '-';
    K = -K;
  }
  if (K >= 100) {
    *p++ = (char)('0' + K / 100);
    K %= 100;
    *p++ = (char)('0' + K / 10);
    *p++ = (char)('0' + K % 10);
  } else if (K >= 10) {
    *p++ = (char)('0' + K / 10);
    *p++ = (char)('0' + K % 10);
  } else
    *p++ = (char)('0' + K);
  return p;
}
inline std::string prettify(const char* digits, int length, int k) {
  const int kk = length + k;  // 10^(kk-1) <= v < 10^kk
  std::string s;
  if (0 <= k && kk <= 21) {  // 1234e7 -> 12340000000.0
    s.assign(digits, (size_t)length);
    s.append((size_t)k, '0');
    s += ".0";
  } else if (0 < kk && kk <= 21) {  // 1234e-2 -> 12.34
    s.assign(digits, (size_t)kk);
    s += '.';
    s.append(digits + kk, (size_t)(length - kk));
  } else if (-6 < kk && kk <= 0) {  // 1234e-6 -> 0.001234
    s = "0.";
    s.append((size_t)(-kk), '0');
    s.append(digits, (size_t)length);
  } else {  // 1e30, 1234e30 -> 1.234e33
    char ex[8];
    char* e = write_exponent(kk - 1, ex);
    s.assign(digits, 1);
    if (length > 1) {
      s += '.';
      s.append(digits + 1, (size_t)(length - 1));
    }
    s += 'e';
    s.append(ex, (size_t)(e - ex));
  }
  return s;
}
// text of a finite double
inline std::string double_text(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  if ((u << 1) == 0) return (u >> 63) ? "-0.0" : "0.0";
  std::string s;
  if (d < 0) {
    s = "-";
    d = -d;
  }
  char buf[32];
  int len = 0, K = 0;
  grisu2(d, buf, &len, &K);
  return s + prettify(buf, len, K);
}

// ---------------------------------------------------------------- a number copied through from the input file
// The reader's verdict on a JSON number literal: the integer kinds when the literal has neither fraction nor
// exponent and fits (32-bit, then 64-bit, signed when it has a minus sign), else a double — converted with full
// precision (correctly rounded; the reference's vendored reader defaults to kParseFullPrecisionFlag), so the C
// library's strtod gives the same value. Returns the text the writer prints for that value; `ok` = false for a
// literal the reader would refuse (malformed, or a magnitude beyond the double range).
// Text -> double, correctly rounded and INDEPENDENT OF THE PROCESS LOCALE: strtod follows LC_NUMERIC (under a locale with
// a decimal comma "1.5" reads as 1), the reference's reader does not — so the conversion runs in a private "C" locale.
inline double parse_double(const std::string& text) {
  static const locale_t c_loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
  return c_loc ? strtod_l(text.c_str(), nullptr, c_loc) : strtod(text.c_str(), nullptr);
}

inline std::string normalize_number(const std::string& text, bool* ok) {
  *ok = true;
  const char* s = text.c_str();
  const bool minus = *s == '-';
  if (minus) s++;
  auto digit = [](char c) { return c >= '0' && c <= '9'; };
  const char* int_begin = s;
  if (*s == '0')
    s++;
  else if (*s >= '1' && *s <= '9')
    while (digit(*s)) s++;
  else {
    *ok = false;
    return text;
  }
  const char* int_end = s;
  bool is_int = true;
  if (*s == '.') {
    is_int = false;
    s++;
    if (!digit(*s)) {
      *ok = false;
      return text;
    }
    while (digit(*s)) s++;
  }
  if (*s == 'e' || *s == 'E') {
    is_int = false;
    s++;
    if (*s == '+' || *s == '-') s++;
    if (!digit(*s)) {
      *ok = false;
      return text;
    }
    while (digit(*s)) s++;
  }
  if (*s) {
    *ok = false;
    return text;
  }
  if (is_int) {
    // magnitude as an unsigned 64-bit integer if it fits the kind the reader would pick
    const size_t nd = (size_t)(int_end - int_begin);
    bool fits = nd <= 20;
    unsigned __int128 m = 0;
    if (fits)
      for (const char* q = int_begin; q < int_end; q++) m = m * 10 + (unsigned)(*q - '0');
    const unsigned __int128 lim = minus ? ((unsigned __int128)1 << 63) : (((unsigned __int128)1 << 64) - 1);
    if (fits && m <= lim) {
      char buf[32];
      const unsigned long long v = (unsigned long long)m;
      if (minus && v)
        snprintf(buf, sizeof(buf), "-%llu", v);
      else
        snprintf(buf, sizeof(buf), "%llu", v);  // "-0" is the integer 0
      return buf;
    }
  }
  const double d = parse_double(text);
  if (!(d == d) || d > 1.7976931348623157e308 || d < -1.7976931348623157e308) {  // "number too big" for the reader
    *ok = false;
    return text;
  }
  return double_text(d);
}

// ---------------------------------------------------------------- strings
// decoded UTF-8 -> the writer's escaped text (quotes not included)
inline std::string escape_string(const std::string& v) {
  static const char hex[] = "0123456789ABCDEF";
  std::string s;
  for (unsigned char c : v) {
    if (c == '"')
      s += "\\\"";
    else if (c == '\\')
      s += "\\\\";
    else if (c < 0x20) {
      switch (c) {
        case 8: s += "\\b"; break;
        case 9: s += "\\t"; break;
        case 10: s += "\\n"; break;
        case 12: s += "\\f"; break;
        case 13: s += "\\r"; break;
        default:
          s += "\\u00";
          s += hex[c >> 4];
          s += hex[c & 15];
      }
    } else
      s += (char)c;
  }
  return s;
}
inline void append_utf8(std::string& s, uint32_t cp) {
  if (cp < 0x80)
    s += (char)cp;
  else if (cp < 0x800) {
    s += (char)(0xC0 | (cp >> 6));
    s += (char)(0x80 | (cp & 0x3F));
  } else if (cp < 0x10000) {
    s += (char)(0xE0 | (cp >> 12));
    s += (char)(0x80 | ((cp >> 6) & 0x3F));
    s += (char)(0x80 | (cp & 0x3F));
  } else {
    s += (char)(0xF0 | (cp >> 18));
    s += (char)(0x80 | ((cp >> 12) & 0x3F));
    s += (char)(0x80 | ((cp >> 6) & 0x3F));
    s += (char)(0x80 | (cp & 0x3F));
  }
}

}  // namespace eg3d_json
