// frame_writer.h -- the per-slice frame of `--img` / `--video` (reference: DVS_flow::recompute, dvs_flow.h:256-335).
//
// The reference composes, through OpenCV, a 2 x 2 mosaic per slice
//     events as recorded (grey)      | their colour-coded time image
//     motion compensated (grey)      | its colour-coded time image
// with every tile resized to (RES_Y * 3) x (RES_X * 3), overlays slice statistics with cv::putText, and writes a
// JPEG (`--img`) or appends to a cv::VideoWriter (`--video`).  The four tiles come from the device
// (bf_projection_img, bf_color_time_img).  What this build does differently -- none of it is on the motion-
// compensation path, all of it is OpenCV's arithmetic / codecs in the reference (un-versioned: parity unpinned):
//   * resize: bilinear with half-pixel centres in float, rounded to nearest (cv::resize INTER_LINEAR uses 11-bit
//     fixed-point coefficients);
//   * no text overlay (Hershey font tables are OpenCV data): the same lines go to a side-car `frame_N.txt`;
//   * containers: binary PPM instead of JPEG, uncompressed AVI ('DIB ', 24-bit bottom-up BGR, with an idx1 index;
//     RIFF's 4 GB limit closes the file early with a warning) instead of a compressed stream.
#ifndef BF_HOST_FRAME_WRITER_H
#define BF_HOST_FRAME_WRITER_H

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace bf {

struct FrameBGR {
    int rows = 0, cols = 0;
    std::vector<uint8_t> px;   // rows x cols x 3, B G R
    FrameBGR() {}
    FrameBGR(int r, int c) : rows(r), cols(c), px((size_t)r * (size_t)c * 3) {}
    uint8_t *at(int r, int c) { return &px[((size_t)r * cols + c) * 3]; }
    const uint8_t *at(int r, int c) const { return &px[((size_t)r * cols + c) * 3]; }
};

// cv::cvtColor(GRAY2RGB): the grey value in all three channels
inline FrameBGR gray_to_bgr(const uint8_t *g, int rows, int cols) {
    FrameBGR f(rows, cols);
    for (size_t i = 0; i < (size_t)rows * cols; ++i) f.px[3 * i] = f.px[3 * i + 1] = f.px[3 * i + 2] = g[i];
    return f;
}

// cv::resize(src, dst, Size(cols, rows)) with INTER_LINEAR: sample position (d + 0.5) * (src / dst) - 0.5, clamped
inline FrameBGR resize_bilinear(const FrameBGR &src, int rows, int cols) {
    if (rows == src.rows && cols == src.cols) return src;
    FrameBGR dst(rows, cols);
    const float sr = (float)src.rows / (float)rows, sc = (float)src.cols / (float)cols;
    std::vector<int> c0(cols), c1(cols);
    std::vector<float> wc(cols);
    for (int c = 0; c < cols; ++c) {
        float x = ((float)c + 0.5f) * sc - 0.5f;
        if (x < 0) x = 0;
        int i = (int)std::floor(x);
        if (i > src.cols - 1) i = src.cols - 1;
        c0[c] = i; c1[c] = i + 1 < src.cols ? i + 1 : src.cols - 1; wc[c] = x - (float)i;
    }
    for (int r = 0; r < rows; ++r) {
        float y = ((float)r + 0.5f) * sr - 0.5f;
        if (y < 0) y = 0;
        int i = (int)std::floor(y);
        if (i > src.rows - 1) i = src.rows - 1;
        const int r0 = i, r1 = i + 1 < src.rows ? i + 1 : src.rows - 1;
        const float wr = y - (float)i;
        for (int c = 0; c < cols; ++c) {
            const uint8_t *p00 = src.at(r0, c0[c]), *p01 = src.at(r0, c1[c]), *p10 = src.at(r1, c0[c]), *p11 = src.at(r1, c1[c]);
            uint8_t *o = dst.at(r, c);
            for (int k = 0; k < 3; ++k) {
                const float top = (float)p00[k] + ((float)p01[k] - (float)p00[k]) * wc[c];
                const float bot = (float)p10[k] + ((float)p11[k] - (float)p10[k]) * wc[c];
                const float v = top + (bot - top) * wr;
                o[k] = (uint8_t)(v <= 0.f ? 0 : (v >= 255.f ? 255 : (int)std::lrint(v)));
            }
        }
    }
    return dst;
}

// cv::hconcat / cv::vconcat of four equally sized tiles (dvs_flow.h:317-323)
inline FrameBGR mosaic_2x2(const FrameBGR &tl, const FrameBGR &tr, const FrameBGR &bl, const FrameBGR &br) {
    FrameBGR m(tl.rows * 2, tl.cols * 2);
    const FrameBGR *t[2][2] = {{&tl, &tr}, {&bl, &br}};
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < tl.rows; ++r)
                std::memcpy(m.at(a * tl.rows + r, b * tl.cols), t[a][b]->at(r, 0), (size_t)tl.cols * 3);
    return m;
}

inline bool write_ppm(const std::string &path, const FrameBGR &f) {
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) return false;
    std::fprintf(fp, "P6\n%d %d\n255\n", f.cols, f.rows);
    std::vector<uint8_t> row((size_t)f.cols * 3);
    for (int r = 0; r < f.rows; ++r) {
        const uint8_t *s = f.at(r, 0);
        for (int c = 0; c < f.cols; ++c) { row[3 * c] = s[3 * c + 2]; row[3 * c + 1] = s[3 * c + 1]; row[3 * c + 2] = s[3 * c]; }
        std::fwrite(row.data(), 1, row.size(), fp);
    }
    return std::fclose(fp) == 0;
}

// Uncompressed AVI 1.0 writer: RIFF 'AVI ' { LIST hdrl { avih, LIST strl { strh, strf } }, LIST movi { 00db ... }, idx1 }
class AviWriter {
    FILE *fp = nullptr;
    int rows = 0, cols = 0, fps = 30;
    uint32_t frames = 0, stride = 0, frame_bytes = 0;
    long movi_pos = 0;
    bool full = false;
    std::vector<uint8_t> buf;

    void u32(uint32_t v) { uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; std::fwrite(b, 1, 4, fp); }
    void u16(uint16_t v) { uint8_t b[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; std::fwrite(b, 1, 2, fp); }
    void tag(const char *t) { std::fwrite(t, 1, 4, fp); }

    void header() {
        std::fseek(fp, 0, SEEK_SET);
        const uint32_t movi_size = 4 + frames * (8 + frame_bytes), idx_size = frames * 16;
        tag("RIFF"); u32(4 + (12 + 64 + 12 + 64 + 48) + (8 + movi_size) + (8 + idx_size)); tag("AVI ");
        tag("LIST"); u32(4 + 64 + 12 + 64 + 48); tag("hdrl");
        tag("avih"); u32(56);
        u32(1000000u / (uint32_t)fps); u32(frame_bytes * (uint32_t)fps); u32(0); u32(0x10 /* has index */);
        u32(frames); u32(0); u32(1); u32(frame_bytes); u32((uint32_t)cols); u32((uint32_t)rows);
        u32(0); u32(0); u32(0); u32(0);
        tag("LIST"); u32(4 + 64 + 48); tag("strl");
        tag("strh"); u32(56);
        tag("vids"); tag("DIB "); u32(0); u16(0); u16(0); u32(0); u32(1); u32((uint32_t)fps); u32(0); u32(frames);
        u32(frame_bytes); u32(0xffffffffu); u32(0); u16(0); u16(0); u16((uint16_t)cols); u16((uint16_t)rows);
        tag("strf"); u32(40);
        u32(40); u32((uint32_t)cols); u32((uint32_t)rows); u16(1); u16(24); u32(0); u32(frame_bytes); u32(0); u32(0); u32(0); u32(0);
        tag("LIST"); u32(movi_size); tag("movi");
    }

public:
    ~AviWriter() { close(); }
    bool is_open() const { return fp != nullptr; }
    uint32_t frame_count() const { return frames; }

    bool open(const std::string &path, int rows_, int cols_, int fps_) {
        close();
        fp = std::fopen(path.c_str(), "wb");
        if (!fp) return false;
        rows = rows_; cols = cols_; fps = fps_ > 0 ? fps_ : 30;
        stride = ((uint32_t)cols * 3 + 3) & ~3u;
        frame_bytes = stride * (uint32_t)rows;
        frames = 0; full = false;
        buf.assign(frame_bytes, 0);
        header();
        movi_pos = std::ftell(fp);
        return true;
    }

    bool write(const FrameBGR &f) {
        if (!fp || f.rows != rows || f.cols != cols) return false;
        // RIFF sizes are 32 bit: stop before the file would pass 4 GB
        if ((uint64_t)(frames + 1) * (frame_bytes + 8 + 16) + 4096 > 0xffffffffull) {
            if (!full) std::fprintf(stderr, "AviWriter: 4 GB container limit reached after %u frames\n", frames);
            full = true;
            return false;
        }
        for (int r = 0; r < rows; ++r) std::memcpy(&buf[(size_t)(rows - 1 - r) * stride], f.at(r, 0), (size_t)cols * 3);
        tag("00db"); u32(frame_bytes);
        std::fwrite(buf.data(), 1, buf.size(), fp);
        frames++;
        return true;
    }

    void close() {
        if (!fp) return;
        tag("idx1"); u32(frames * 16);
        for (uint32_t i = 0; i < frames; ++i) { tag("00db"); u32(0x10); u32(4 + i * (8 + frame_bytes)); u32(frame_bytes); }
        header();
        std::fclose(fp);
        fp = nullptr;
    }
};

}  // namespace bf

#endif  // BF_HOST_FRAME_WRITER_H
