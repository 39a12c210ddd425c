// event_reader.h -- event input for the CLI and EventFile::from_file (the reference's reader is the
// `ifstream >> double >> uint >> uint >> bool` loop of bf_motion_compensator.cpp:179-206 and
// event_file.h:141-176).  Two formats, detected from the first bytes:
//
//   text    "t x y p" per line (seconds, column, row, polarity).  Same values as the iostream loop:
//           the time is converted by the exact fast path (decimal mantissa < 2^53 and a power of ten
//           <= 10^22: one correctly rounded division, Clinger 1990) or by strtod, the integers by a
//           digit loop, and reading stops at the first malformed record as `>>` would.  The whole
//           file is read in one piece: at >= 1 Gev/s on the device the locale-aware iostream
//           extraction (~0.4 us per value) was the slowest stage of the CLI.
//   binary  structure-of-arrays, little endian: char magic[8] = "BFEVSOA1", u64 n, then u64 t_ns[n]
//           (absolute), u16 x[n] (column), u16 y[n] (row), u8 p[n].  No parsing, 13 B per event.
#ifndef BF_HOST_EVENT_READER_H
#define BF_HOST_EVENT_READER_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace bf {

class EventReader {
    std::vector<char> buf;
    const char *cur = nullptr, *end = nullptr;
    bool binary = false, ok = false;
    uint64_t n_bin = 0, i_bin = 0;
    const uint64_t *b_t = nullptr;
    const uint16_t *b_x = nullptr, *b_y = nullptr;
    const uint8_t *b_p = nullptr;

    static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
    void skip_space() { while (cur < end && is_space(*cur)) ++cur; }

    bool parse_double(double &out) {
        skip_space();
        if (cur >= end) return false;
        const char *s = cur;
        bool neg = false;
        if (*s == '-' || *s == '+') { neg = (*s == '-'); ++s; }
        uint64_t mant = 0;
        int digits = 0, frac = 0;
        bool any = false, simple = true;
        while (s < end && *s >= '0' && *s <= '9') { if (digits < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); ++digits; } else simple = false; ++s; any = true; }
        if (s < end && *s == '.') {
            ++s;
            while (s < end && *s >= '0' && *s <= '9') { if (digits < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); ++digits; ++frac; } else simple = false; ++s; any = true; }
        }
        if (!any) return false;
        if (s < end && !is_space(*s)) simple = false;   // exponent, hex, inf / nan, trailing junk: strtod decides
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                     1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        if (simple && mant < (1ull << 53) && frac <= 22) {
            const double v = (double)mant / p10[frac];   // both operands exact -> one correctly rounded operation
            out = neg ? -v : v;
            cur = s;
            return true;
        }
        char *e = nullptr;
        std::string tok(cur, (size_t)((s - cur) + 64 < end - cur ? (s - cur) + 64 : end - cur));
        out = std::strtod(tok.c_str(), &e);
        if (e == tok.c_str()) return false;
        cur += (e - tok.c_str());
        return true;
    }

    bool parse_uint(unsigned &out) {
        skip_space();
        if (cur >= end || *cur < '0' || *cur > '9') return false;
        uint64_t v = 0;
        while (cur < end && *cur >= '0' && *cur <= '9') { v = v * 10 + (uint64_t)(*cur - '0'); if (v > 0xffffffffull) return false; ++cur; }
        out = (unsigned)v;
        return true;
    }

    bool parse_bool(bool &out) {   // operator>>(bool&) without boolalpha: 0 or 1
        unsigned v;
        if (!parse_uint(v) || v > 1) return false;
        out = v != 0;
        return true;
    }

public:
    explicit EventReader(const std::string &path) {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) return;
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
        std::fclose(f);
        if (got != buf.size()) return;
        ok = true;
        cur = buf.data();
        end = cur + buf.size();
        if (buf.size() >= 16 && std::memcmp(buf.data(), "BFEVSOA1", 8) == 0) {
            binary = true;
            std::memcpy(&n_bin, buf.data() + 8, 8);
            // (the count comes from the file: compare it with the size before multiplying -- a crafted 64-bit count
            // must not wrap the size check)
            if (n_bin > (buf.size() - 16) / 13) { n_bin = 0; ok = false; return; }
            b_t = reinterpret_cast<const uint64_t *>(buf.data() + 16);
            b_x = reinterpret_cast<const uint16_t *>(buf.data() + 16 + n_bin * 8);
            b_y = b_x + n_bin;
            b_p = reinterpret_cast<const uint8_t *>(b_y + n_bin);
        }
    }

    bool good() const { return ok; }
    bool is_binary() const { return binary; }

    // Text records: next value quadruple, false at end of input or at the first malformed record.
    bool next_text(double &t, unsigned &x, unsigned &y, bool &p) {
        return !binary && ok && parse_double(t) && parse_uint(x) && parse_uint(y) && parse_bool(p);
    }
    // Binary records: absolute nanoseconds.
    bool next_binary(uint64_t &t_ns, unsigned &x, unsigned &y, bool &p) {
        if (!binary || i_bin >= n_bin) return false;
        t_ns = b_t[i_bin]; x = b_x[i_bin]; y = b_y[i_bin]; p = b_p[i_bin] != 0;
        ++i_bin;
        return true;
    }

    // The reference's loop (bf_motion_compensator.cpp:186-197): the first record defines t_0, every event
    // gets FROM_SEC(t - t_0) (double arithmetic, truncated); x / y swapped on the way in.  fn(row, col, t_ns).
    template <class F> unsigned long long for_each_event(F fn) {
        unsigned long long cnt = 0;
        unsigned x = 0, y = 0;
        bool p = false;
        if (binary) {
            uint64_t t = 0, t0 = 0;
            if (next_binary(t0, x, y, p)) { fn(y, x, (unsigned long long)0); ++cnt; }
            while (next_binary(t, x, y, p)) { fn(y, x, (unsigned long long)(t - t0)); ++cnt; }
        } else {
            double t = 0, t_0 = 0;
            if (next_text(t_0, x, y, p)) { fn(y, x, (unsigned long long)(1000000000 * (0))); ++cnt; }
            while (next_text(t, x, y, p)) {
                t -= t_0;
                fn(y, x, (unsigned long long)(1000000000 * (t)));
                ++cnt;
            }
        }
        return cnt;
    }

    // Writes the binary form.  t_ns absolute, x = column, y = row.
    static bool write_binary(const std::string &path, const std::vector<uint64_t> &t_ns, const std::vector<uint16_t> &x,
                             const std::vector<uint16_t> &y, const std::vector<uint8_t> &p) {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        const uint64_t n = t_ns.size();
        bool w = std::fwrite("BFEVSOA1", 1, 8, f) == 8 && std::fwrite(&n, 8, 1, f) == 1;
        w = w && (n == 0 || (std::fwrite(t_ns.data(), 8, n, f) == n && std::fwrite(x.data(), 2, n, f) == n &&
                             std::fwrite(y.data(), 2, n, f) == n && std::fwrite(p.data(), 1, n, f) == n));
        return std::fclose(f) == 0 && w;
    }
};

}  // namespace bf

#endif  // BF_HOST_EVENT_READER_H
