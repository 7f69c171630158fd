/* avf_bgzf.h — one BGZF block (SAM spec section 4.1): a gzip member of at most 64 KiB with a BC extra field */
#ifndef AVF_BGZF_H
#define AVF_BGZF_H
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace avf_bgzf {

constexpr size_t kBlock = 0xff00; /* uncompressed bytes per block */

/* every thread keeps one deflate state and one output buffer for all the blocks it compresses: a fresh deflateInit2 per 64 KiB block
 * is two large allocations (memory maps) per block, which many compressing threads serialise on */
struct Deflater {
    z_stream zs;
    bool ok = false;
    std::vector<unsigned char> buf;
    Deflater() : buf(kBlock + 1024) {
        memset(&zs, 0, sizeof(zs));
        ok = deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
    }
    ~Deflater() {
        if (ok) deflateEnd(&zs);
    }
};
inline bool compress_block(const char *in, size_t n, std::string &out) {
    static thread_local Deflater d;
    if (!d.ok || deflateReset(&d.zs) != Z_OK) return false;
    std::vector<unsigned char> &buf = d.buf;
    z_stream &zs = d.zs;
    zs.next_in = (Bytef *)in;
    zs.avail_in = (uInt)n;
    zs.next_out = buf.data() + 18;
    zs.avail_out = (uInt)(buf.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    if (rc != Z_STREAM_END) return false;
    const size_t total = 18 + clen + 8;
    static const unsigned char head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00};
    memcpy(buf.data(), head, 16);
    buf[16] = (unsigned char)((total - 1) & 0xff);
    buf[17] = (unsigned char)((total - 1) >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)in, (uInt)n);
    const uint32_t isize = (uint32_t)n;
    for (int k = 0; k < 4; ++k) {
        buf[18 + clen + k] = (unsigned char)(crc >> (8 * k));
        buf[18 + clen + 4 + k] = (unsigned char)(isize >> (8 * k));
    }
    out.assign((const char *)buf.data(), total);
    return true;
}

inline bool write_eof(FILE *fp) {
    static const unsigned char eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};
    return fwrite(eof, 1, sizeof(eof), fp) == sizeof(eof);
}

/* streaming writer: blocks are compressed and written as they fill */
class Stream {
  public:
    explicit Stream(FILE *fp) : fp_(fp) { cur_.reserve(kBlock); }
    bool write(const char *p, size_t n) {
        while (n) {
            const size_t room = kBlock - cur_.size();
            const size_t take = n < room ? n : room;
            cur_.append(p, take);
            p += take;
            n -= take;
            if (cur_.size() == kBlock && !flush()) return false;
        }
        return true;
    }
    bool write(const std::string &s) { return write(s.data(), s.size()); }
    bool finish() { return (cur_.empty() || flush()) && write_eof(fp_); }

  private:
    bool flush() {
        std::string out;
        if (!compress_block(cur_.data(), cur_.size(), out)) return false;
        cur_.clear();
        return fwrite(out.data(), 1, out.size(), fp_) == out.size();
    }
    FILE *fp_;
    std::string cur_;
};

} // namespace avf_bgzf
#endif
