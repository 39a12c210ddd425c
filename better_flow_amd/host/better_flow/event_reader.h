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
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

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

    // One chunk of a "regular" text file -- every line holds exactly one record or nothing -- parsed on its own.
    struct Chunk {
        const char *from = nullptr, *to = nullptr;
        std::vector<unsigned long long> t_ns;
        std::vector<uint32_t> row, col;
        bool regular = true;
    };
    static void parse_chunk(Chunk &c, double t_0) {
        EventReader r;                 // a cursor over the chunk (no buffer of its own)
        r.cur = c.from; r.end = c.to; r.ok = true;
        while (r.cur < r.end) {
            // a line: spaces, then nothing or exactly "t x y p", then spaces up to the newline
            const char *eol = r.cur;
            while (eol < r.end && *eol != '\n') ++eol;
            const char *save_end = r.end;
            r.end = eol;
            r.skip_space();
            if (r.cur < r.end) {
                double t; unsigned x, y; bool p;
                if (!(r.parse_double(t) && r.parse_uint(x) && r.parse_uint(y) && r.parse_bool(p))) { c.regular = false; return; }
                r.skip_space();
                if (r.cur != r.end) { c.regular = false; return; }
                t -= t_0;
                c.t_ns.push_back((unsigned long long)(1000000000 * (t)));
                c.row.push_back(y); c.col.push_back(x);
            }
            r.end = save_end;
            r.cur = eol < r.end ? eol + 1 : r.end;
        }
    }
    EventReader() {}

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

    // The whole text input at once, on `threads` threads: row / column / FROM_SEC(t - t_0) of every record, exactly what
    // for_each_event() would deliver.  Only for REGULAR files (one record per line, blank lines allowed); returns false --
    // and delivers nothing -- for anything else (a record spread over lines, a malformed record, trailing text), which
    // the caller then reads with for_each_event(), the sequential parser with the iostream loop's exact stopping rule.
    bool parse_text_parallel(int threads, std::vector<unsigned long long> &t_ns, std::vector<uint32_t> &row, std::vector<uint32_t> &col) {
        if (binary || !ok) return false;
        const char *keep = cur;
        double t_0 = 0;
        unsigned x0, y0; bool p0;
        if (!(parse_double(t_0) && parse_uint(x0) && parse_uint(y0) && parse_bool(p0))) { cur = keep; return false; }
        cur = keep;
        if (threads < 1) threads = 1;
        const size_t total = (size_t)(end - cur);
        if ((size_t)threads > total / 65536 + 1) threads = (int)(total / 65536 + 1);
        std::vector<Chunk> chunks((size_t)threads);
        const char *at = cur;
        for (int k = 0; k < threads; ++k) {
            chunks[k].from = at;
            const char *to = (k + 1 == threads) ? end : cur + total * (size_t)(k + 1) / (size_t)threads;
            if (to < at) to = at;
            while (to < end && *to != '\n') ++to;     // chunks end behind a newline
            if (to < end) ++to;
            chunks[k].to = at = to;
        }
        std::vector<std::thread> pool;
        for (int k = 1; k < threads; ++k) pool.emplace_back([&chunks, k, t_0] { parse_chunk(chunks[k], t_0); });
        parse_chunk(chunks[0], t_0);
        for (auto &th : pool) th.join();
        size_t n = 0;
        for (const Chunk &c : chunks) { if (!c.regular) return false; n += c.t_ns.size(); }
        t_ns.resize(n); row.resize(n); col.resize(n);
        size_t o = 0;
        for (const Chunk &c : chunks) {
            std::memcpy(t_ns.data() + o, c.t_ns.data(), c.t_ns.size() * 8);
            std::memcpy(row.data() + o, c.row.data(), c.row.size() * 4);
            std::memcpy(col.data() + o, c.col.data(), c.col.size() * 4);
            o += c.t_ns.size();
        }
        cur = end;
        return true;
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

// The binary event file read column block by column block, without a copy of the whole file in memory: pread()
// straight into the caller's arrays (for StreamEngine: into the pinned ring itself), on several threads.
class SoaFile {
    int fd = -1;
    uint64_t n_ = 0;
    bool ok_ = false;

public:
    explicit SoaFile(const std::string &path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        char head[16];
        const off_t size = ::lseek(fd, 0, SEEK_END);
        if (size < 16 || ::pread(fd, head, 16, 0) != 16 || std::memcmp(head, "BFEVSOA1", 8) != 0) return;
        std::memcpy(&n_, head + 8, 8);
        if (n_ > ((uint64_t)size - 16) / 13) { n_ = 0; return; }   // (compare before multiplying: the count comes from the file)
        ok_ = true;
    }
    ~SoaFile() { if (fd >= 0) ::close(fd); }
    SoaFile(const SoaFile &) = delete;
    SoaFile &operator=(const SoaFile &) = delete;

    bool good() const { return ok_; }
    uint64_t events() const { return n_; }
    static bool is_soa(const std::string &path) {
        char magic[8];
        FILE *f = std::fopen(path.c_str(), "rb");
        const bool yes = f && std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "BFEVSOA1", 8) == 0;
        if (f) std::fclose(f);
        return yes;
    }

    // events [first, first + n): absolute timestamps, rows (the file's y column), columns (its x column)
    bool read(uint64_t first, uint64_t n, uint64_t *t_ns, uint16_t *row, uint16_t *col) const {
        auto all = [this](void *dst, uint64_t bytes, uint64_t off) {
            char *d = (char *)dst;
            while (bytes > 0) {
                const ssize_t got = ::pread(fd, d, bytes, (off_t)off);
                if (got <= 0) return false;
                d += got; off += (uint64_t)got; bytes -= (uint64_t)got;
            }
            return true;
        };
        return all(t_ns, n * 8, 16 + first * 8) && all(col, n * 2, 16 + n_ * 8 + first * 2) &&
               all(row, n * 2, 16 + n_ * 10 + first * 2);
    }
};

// Feed a binary event file to a StreamEngine-like sink (reserve / commit / set_time_base) block by block, the blocks
// read into the sink's own ring on `threads` threads.  Same event stream as EventReader::for_each_event: the first
// record's time is the origin.  Returns the number of events fed; *ok = false on a read error.
template <class Sink> uint64_t feed_soa_file(const SoaFile &file, Sink &sink, int threads, bool *ok, double *read_seconds = nullptr) {
    *ok = file.good();
    if (!file.good() || file.events() == 0) return 0;
    uint64_t t0 = 0;
    uint16_t r0, c0;
    if (!file.read(0, 1, &t0, &r0, &c0)) { *ok = false; return 0; }
    sink.set_time_base(t0);
    if (threads < 1) threads = 1;
    uint64_t done = 0;
    while (done < file.events()) {
        typename Sink::Span sp[2];
        const uint64_t want = file.events() - done;
        // (blocks of a million events: large enough to amortise the reader threads, small enough for the slices they close
        // to reach the solver while the next block is being read)
        const size_t got = sink.reserve(want > (uint64_t)1 << 20 ? (size_t)1 << 20 : (size_t)want, sp);
        // split the granted slots into per-thread pieces of at least 64k events
        struct Piece { uint64_t first, n; uint64_t *t; uint16_t *r, *c; };
        std::vector<Piece> pieces;
        uint64_t at = done;
        for (int p = 0; p < 2; ++p) {
            const size_t per = sp[p].n / (size_t)threads > 65536 ? sp[p].n / (size_t)threads + 1 : 65536;
            for (size_t o = 0; o < sp[p].n; o += per) {
                const size_t m = sp[p].n - o < per ? sp[p].n - o : per;
                pieces.push_back(Piece{at + o, m, sp[p].timestamp + o, sp[p].row + o, sp[p].col + o});
            }
            at += sp[p].n;
        }
        const auto t_read = std::chrono::steady_clock::now();
        std::vector<char> fine(pieces.size(), 1);
        std::vector<std::thread> pool;
        for (size_t k = 1; k < pieces.size(); ++k)
            pool.emplace_back([&, k] { fine[k] = file.read(pieces[k].first, pieces[k].n, pieces[k].t, pieces[k].r, pieces[k].c) ? 1 : 0; });
        if (!pieces.empty()) fine[0] = file.read(pieces[0].first, pieces[0].n, pieces[0].t, pieces[0].r, pieces[0].c) ? 1 : 0;
        for (auto &th : pool) th.join();
        if (read_seconds) *read_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_read).count();
        for (char f : fine) if (!f) { *ok = false; return done; }
        sink.commit(got);
        done += got;
    }
    return done;
}

}  // namespace bf

#endif  // BF_HOST_EVENT_READER_H
