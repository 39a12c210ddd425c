// event_file.h -- text event I/O (mirror of the I/O half of the reference's
// better_flow/event_file.h:141-176,265-289).  The visualisation half of that file (colour maps,
// flow arrows, OpenCV rendering) is outside the motion-compensation path.
#ifndef BF_HOST_EVENT_FILE_H
#define BF_HOST_EVENT_FILE_H

#include <better_flow/common.h>
#include <better_flow/event.h>

#include <better_flow/event_reader.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <fcntl.h>
#include <unistd.h>

namespace bf {

// "%.9f" of a double without going through printf: the same characters (the exact binary value, rounded to nine
// decimals half-to-even, as glibc does), ~20x faster.  x = m * 2^e with an integer m < 2^53; m * 10^9 fits 128 bits, and
// the shift by -e with an exact remainder gives the correctly rounded count of 1e-9 units.  Values of 2^63 and above,
// infinities and NaNs go to snprintf.  Returns the number of characters written (no terminator).
inline int format_fixed9(double x, char *out) {
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    const int bexp = (int)((bits >> 52) & 0x7ff);
    const uint64_t frac = bits & 0xfffffffffffffull;
    if (bexp == 0x7ff || bexp - 1075 > 10) return std::snprintf(out, 400, "%.9f", x);
    char *p = out;
    if (bits >> 63) *p++ = '-';
    const uint64_t m = bexp ? (frac | (1ull << 52)) : frac;
    const int e = bexp ? bexp - 1075 : -1074;
    unsigned __int128 units;                       // |x| in units of 1e-9, rounded
    if (e >= 0) {
        units = ((unsigned __int128)(m << e)) * 1000000000u;
    } else {
        const unsigned __int128 prod = (unsigned __int128)m * 1000000000u;   // < 2^83
        const int s = -e;
        if (s >= 128) units = 0;                   // |x| * 1e9 < 2^83 / 2^128: far below one half
        else {
            units = prod >> s;
            const unsigned __int128 rem = prod & ((((unsigned __int128)1) << s) - 1), half = ((unsigned __int128)1) << (s - 1);
            if (rem > half || (rem == half && (units & 1))) ++units;
        }
    }
    uint64_t ip = (uint64_t)(units / 1000000000u);
    uint32_t fp = (uint32_t)(units % 1000000000u);
    char tmp[24];
    int nd = 0;
    do { tmp[nd++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (nd) *p++ = tmp[--nd];
    *p++ = '.';
    for (int k = 8; k >= 0; --k) { p[k] = (char)('0' + fp % 10); fp /= 10; }
    p += 9;
    return (int)(p - out);
}

// EventFile::to_file_uv for a structure-of-arrays table (StreamEngine::get_accumulated): "t x y 1 best_v best_u" with
// nine decimals, x / y and u / v swapped back (event_file.h:272-276).  The lines are formatted on `threads` threads,
// one contiguous part of the table each, and written in order.  Returns false if the file cannot be written.
inline bool write_flow_text(const std::string &fname, const std::vector<uint64_t> &ts, const std::vector<uint16_t> &row,
                            const std::vector<uint16_t> &col, const std::vector<double> &u, const std::vector<double> &v, int threads) {
    const size_t n = ts.size();
    if (threads < 1) threads = 1;
    // Chunks of ~128k lines, formatted by `threads` workers (next chunk by an atomic counter) and written IN ORDER by one
    // more thread as they become ready: copying ~50 bytes per line into the page cache takes as long as eight threads
    // take to format them, so the two overlap instead of following each other.
    const size_t chunk_lines = 131072;
    const size_t nchunks = n ? (n + chunk_lines - 1) / chunk_lines : 0;
    if ((size_t)threads > nchunks) threads = nchunks ? (int)nchunks : 1;
    const int fd = ::open(fname.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    std::vector<std::string> parts(nchunks);
    std::vector<char> ready(nchunks, 0);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    auto format_chunk = [&](size_t k) {
        const size_t a = k * chunk_lines, b = std::min(n, a + chunk_lines);
        std::string &out = parts[k];
        out.resize((b - a) * 96 + 64);
        char *p = &out[0];
        const char *lim = p + out.size();
        for (size_t i = a; i < b; ++i) {
            if (lim - p < 1400) {   // three %.9f of huge values could be long: keep room
                const size_t used = (size_t)(p - &out[0]);
                out.resize(out.size() * 2 + 2048);
                p = &out[0] + used;
                lim = &out[0] + out.size();
            }
            p += format_fixed9(double(ts[i]) / 1000000000, p);
            *p++ = ' ';
            unsigned xy[2] = {col[i], row[i]};
            for (unsigned val : xy) {
                char tmp[8];
                int nd = 0;
                do { tmp[nd++] = (char)('0' + val % 10); val /= 10; } while (val);
                while (nd) *p++ = tmp[--nd];
                *p++ = ' ';
            }
            *p++ = '1'; *p++ = ' ';
            p += format_fixed9(v[i], p);
            *p++ = ' ';
            p += format_fixed9(u[i], p);
            *p++ = '\n';
        }
        out.resize((size_t)(p - &out[0]));
    };
    auto work = [&]() {
        for (size_t k = next.fetch_add(1); k < nchunks; k = next.fetch_add(1)) {
            format_chunk(k);
            { std::lock_guard<std::mutex> lk(mu); ready[k] = 1; }
            cv.notify_one();
        }
    };
    bool good = true;
    auto writer = [&]() {
        off_t at = 0;
        for (size_t k = 0; k < nchunks; ++k) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready[k] != 0; }); }
            std::string &part = parts[k];
            size_t done = 0;
            while (good && done < part.size()) {
                const ssize_t w = ::pwrite(fd, part.data() + done, part.size() - done, at + (off_t)done);
                if (w <= 0) good = false;
                else done += (size_t)w;
            }
            at += (off_t)part.size();
            std::string().swap(part);   // (the text of a long recording need not be held twice)
        }
    };
    std::vector<std::thread> pool;
    for (int k = 2; k < threads; ++k) pool.emplace_back(work);   // (`threads` in all: the writer is one of them)
    std::thread wr(writer);
    work();
    for (auto &th : pool) th.join();
    wr.join();
    return ::close(fd) == 0 && good;
}

}  // namespace bf

class EventFile {
public:
    // "t x y p" per line: seconds, column, row, polarity.  x / y are swapped on the way in
    // (event_file.h:160,167; bf_motion_compensator.cpp:192,200): fr_x = row, fr_y = column.
    template <class T> static void from_file(T *events, std::string fname) {
        std::cout << "Reading from file... (" << fname << ")" << std::endl << std::flush;
        clock_t begin = std::clock();
        bf::EventReader reader(fname);   // text (the reference's format) or binary SoA, auto-detected
        ull cnt = reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
            events->push_back(Event(row, col, (ull)t_ns));
        });
        clock_t end = std::clock();
        if (cnt == 0) {
            std::cout << "Read " << cnt << " events, finished" << std::endl << std::endl << std::flush;
            return;
        }
        std::cout << "Read " << cnt << " events, finished" << std::endl << std::flush;
        std::cout << "Elapsed: " << double(end - begin) / CLOCKS_PER_SEC << " sec." << std::endl << std::flush;
    }

    // "t x y 1 best_v best_u", 9 decimals; x / y and u / v swapped back (event_file.h:272-276).
    template <class T> static void to_file_uv(T *events, std::string fname) {
        std::cout << "Writing events and flow to file... (" << fname << ")" << std::endl << std::flush;
        ull cnt = 0;
        clock_t begin = std::clock();
        // same characters as `ofstream << std::fixed << std::setprecision(9)` (event_file.h:272-276), formatted
        // into one buffer instead of one flushed line at a time
        std::string out;
        out.reserve(events->size() * 64 + 64);
        char line[1400];
        for (auto &e : *events) {
            char *p = line;
            p += bf::format_fixed9(double(e.timestamp) / 1000000000, p);
            p += std::snprintf(p, 64, " %u %u 1 ", (unsigned)e.fr_y, (unsigned)e.fr_x);
            p += bf::format_fixed9(e.best_v, p);
            *p++ = ' ';
            p += bf::format_fixed9(e.best_u, p);
            *p++ = '\n';
            out.append(line, (size_t)(p - line));
            cnt++;
        }
        if (FILE *f = std::fopen(fname.c_str(), "wb")) {
            std::fwrite(out.data(), 1, out.size(), f);
            std::fclose(f);
        }
        clock_t end = std::clock();
        if (cnt == 0) {
            std::cout << "Written " << cnt << " events, finished" << std::endl << std::endl << std::flush;
            return;
        }
        std::cout << "Written " << cnt << " events, finished" << std::endl << std::flush;
        std::cout << "Elapsed: " << double(end - begin) / CLOCKS_PER_SEC << " sec." << std::endl << std::flush;
    }
};

#endif  // BF_HOST_EVENT_FILE_H
