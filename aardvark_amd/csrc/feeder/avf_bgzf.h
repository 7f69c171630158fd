/* avf_bgzf.h — one BGZF block (SAM spec section 4.1): a gzip member of at most 64 KiB with a BC extra field */
#ifndef AVF_BGZF_H
#define AVF_BGZF_H
#include <dlfcn.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace avf_bgzf {

constexpr size_t kBlock = 0xff00; /* uncompressed bytes per block */

/* libdeflate, when the system has it (looked up at run time: the image ships libdeflate.so.0 without its header): the same DEFLATE streams at the same level
 * in a third of zlib's time — the annotated VCFs of a whole genome are 300 MB of text, their compression was 0.4 s of the tool's 1.45 s on 16 cores.
 * AVF_ZLIB=1 in the environment keeps zlib.  The prototypes are libdeflate's public C API (libdeflate.h, stable since 1.0). */
struct LibDeflate {
    void *(*alloc_compressor)(int) = nullptr;
    size_t (*deflate_compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*free_compressor)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    void *(*alloc_decompressor)() = nullptr;
    int (*deflate_decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_decompressor)(void *) = nullptr;
    bool ok = false;
    LibDeflate() {
        if (getenv("AVF_ZLIB")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc_compressor = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        deflate_compress = (size_t(*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        free_compressor = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        crc32 = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        alloc_decompressor = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        deflate_decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        free_decompressor = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        ok = alloc_compressor && deflate_compress && free_compressor && crc32 && alloc_decompressor && deflate_decompress && free_decompressor;
    }
};
inline const LibDeflate &libdeflate() {
    static const LibDeflate l;
    return l;
}
/* one raw DEFLATE stream of known uncompressed size into `out` (a BGZF block's payload): libdeflate when present, else zlib on the caller's z_stream (inflateInit2(-15)) */
struct Inflater {
    void *ld = nullptr;
    z_stream zs;
    bool z_ok = false;
    Inflater() {
        memset(&zs, 0, sizeof(zs));
        if (libdeflate().ok) ld = libdeflate().alloc_decompressor();
        if (!ld) z_ok = inflateInit2(&zs, -15) == Z_OK;
    }
    ~Inflater() {
        if (ld) libdeflate().free_decompressor(ld);
        if (z_ok) inflateEnd(&zs);
    }
    Inflater(const Inflater &) = delete;
    Inflater &operator=(const Inflater &) = delete;
    bool ok() const { return ld || z_ok; }
    bool run(const void *in, size_t n_in, void *out, size_t n_out) {
        if (ld) {
            size_t got = 0;
            return libdeflate().deflate_decompress(ld, in, n_in, out, n_out, &got) == 0 && got == n_out;
        }
        if (!z_ok || inflateReset(&zs) != Z_OK) return false;
        zs.next_in = (Bytef *)in;
        zs.avail_in = (uInt)n_in;
        zs.next_out = (Bytef *)out;
        zs.avail_out = (uInt)n_out;
        return inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
    }
};

/* every thread keeps one deflate state and one output buffer for all the blocks it compresses: a fresh deflateInit2 per 64 KiB block
 * is two large allocations (memory maps) per block, which many compressing threads serialise on */
struct Deflater {
    z_stream zs;
    bool ok = false;
    void *ld = nullptr;
    std::vector<unsigned char> buf;
    Deflater() : buf(kBlock + 1024) {
        memset(&zs, 0, sizeof(zs));
        if (libdeflate().ok) ld = libdeflate().alloc_compressor(6);
        if (!ld) ok = deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
    }
    ~Deflater() {
        if (ld) libdeflate().free_compressor(ld);
        if (ok) deflateEnd(&zs);
    }
};
inline bool compress_block(const char *in, size_t n, std::string &out) {
    static thread_local Deflater d;
    std::vector<unsigned char> &buf = d.buf;
    size_t clen = 0;
    uint32_t crc = 0;
    if (d.ld) {
        clen = libdeflate().deflate_compress(d.ld, in, n, buf.data() + 18, buf.size() - 18 - 8);
        if (clen == 0) return false; /* (does not fit: text never grows by a kilobyte per block) */
        crc = libdeflate().crc32(0, in, n);
    } else {
        if (!d.ok || deflateReset(&d.zs) != Z_OK) return false;
        z_stream &zs = d.zs;
        zs.next_in = (Bytef *)in;
        zs.avail_in = (uInt)n;
        zs.next_out = buf.data() + 18;
        zs.avail_out = (uInt)(buf.size() - 18 - 8);
        const int rc = deflate(&zs, Z_FINISH);
        clen = zs.total_out;
        if (rc != Z_STREAM_END) return false;
        crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)in, (uInt)n);
    }
    const size_t total = 18 + clen + 8;
    static const unsigned char head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00};
    memcpy(buf.data(), head, 16);
    buf[16] = (unsigned char)((total - 1) & 0xff);
    buf[17] = (unsigned char)((total - 1) >> 8);
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
