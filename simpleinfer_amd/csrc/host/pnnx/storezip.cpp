#include "storezip.h"

#include <cstdint>
#include <cstdio>
#include <cstring>

namespace pnnx {

namespace {
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
}  // namespace

int StoreZipReader::open(const std::string& path) {
    close();
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) {
        fprintf(stderr, "storezip: cannot open %s\n", path.c_str());
        return -1;
    }
    fseek(fp, 0, SEEK_END);
    const long len = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    blob_.resize(len > 0 ? (size_t)len : 0);
    const size_t got = blob_.empty() ? 0 : fread(blob_.data(), 1, blob_.size(), fp);
    fclose(fp);
    if (got != blob_.size()) {
        fprintf(stderr, "storezip: short read on %s\n", path.c_str());
        return -1;
    }

    // records follow one another from offset 0: local file headers, then the central directory
    size_t pos = 0;
    const size_t n = blob_.size();
    while (pos + 4 <= n) {
        const uint32_t sig = rd32(&blob_[pos]);
        if (sig == 0x04034b50u) {  // local file header, 30 fixed bytes
            if (pos + 30 > n) return -1;
            const unsigned char* h = &blob_[pos];
            const uint16_t flag = rd16(h + 6), method = rd16(h + 8);
            const uint32_t csize = rd32(h + 18), usize = rd32(h + 22);
            const uint16_t name_len = rd16(h + 26), extra_len = rd16(h + 28);
            if (flag & 0x08) {
                fprintf(stderr, "storezip: data descriptors are not supported\n");
                return -1;
            }
            if (method != 0 || csize != usize) {
                fprintf(stderr, "storezip: entry is not stored (method %u)\n", method);
                return -1;
            }
            const size_t data_off = pos + 30 + name_len + extra_len;
            if (data_off + csize > n) return -1;
            std::string name(reinterpret_cast<const char*>(h + 30), name_len);
            Entry e;
            e.offset = data_off;
            e.size = csize;
            entries_[name] = e;
            pos = data_off + csize;
        } else if (sig == 0x02014b50u) {  // central directory header, 46 fixed bytes
            if (pos + 46 > n) return -1;
            const unsigned char* h = &blob_[pos];
            pos += 46 + (size_t)rd16(h + 28) + rd16(h + 30) + rd16(h + 32);
        } else if (sig == 0x06054b50u) {  // end of central directory, 22 fixed bytes
            if (pos + 22 > n) return -1;
            pos += 22 + (size_t)rd16(&blob_[pos + 20]);
        } else {
            fprintf(stderr, "storezip: unsupported record signature %08x\n", sig);
            return -1;
        }
    }
    return 0;
}

size_t StoreZipReader::get_file_size(const std::string& name) const {
    auto it = entries_.find(name);
    return it == entries_.end() ? 0 : it->second.size;
}

int StoreZipReader::read_file(const std::string& name, char* data) const {
    auto it = entries_.find(name);
    if (it == entries_.end()) return -1;
    if (it->second.size) memcpy(data, &blob_[it->second.offset], it->second.size);
    return 0;
}

void StoreZipReader::close() {
    blob_.clear();
    blob_.shrink_to_fit();
    entries_.clear();
}

// ---- writer ---------------------------------------------------------------------------------------
namespace {

uint32_t crc32_of(const char* data, size_t size) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            table[n] = c;
        }
        ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < size; ++i) c = table[(c ^ (unsigned char)data[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

struct ByteSink {
    std::vector<unsigned char> b;
    void u16(uint32_t v) { b.push_back((unsigned char)(v & 0xFF)); b.push_back((unsigned char)((v >> 8) & 0xFF)); }
    void u32(uint32_t v) { u16(v & 0xFFFFu); u16(v >> 16); }
    void str(const std::string& s) { b.insert(b.end(), s.begin(), s.end()); }
};

}  // namespace

StoreZipWriter::~StoreZipWriter() {
    if (fp_) close();
}

int StoreZipWriter::open(const std::string& path) {
    if (fp_) close();
    records_.clear();
    cursor_ = 0;
    failed_ = false;
    fp_ = fopen(path.c_str(), "wb");
    if (!fp_) {
        fprintf(stderr, "storezip: cannot create %s\n", path.c_str());
        return -1;
    }
    return 0;
}

int StoreZipWriter::write_file(const std::string& name, const char* data, size_t size) {
    if (!fp_ || name.empty() || name.size() > 0xFFFFu) return -1;
    if (size > 0xFFFFFFFFull || cursor_ + 30 + name.size() + size > 0xFFFFFFFFull) {
        fprintf(stderr, "storezip: %s does not fit a non-zip64 archive\n", name.c_str());
        return -1;
    }
    Record r;
    r.name = name;
    r.crc = crc32_of(data, size);
    r.size = (unsigned int)size;
    r.offset = (unsigned int)cursor_;

    ByteSink h;  // local file header
    h.u32(0x04034b50u);
    h.u16(10);      // version needed: 1.0 (stored)
    h.u16(0);       // flags: sizes and crc are in this header, no data descriptor
    h.u16(0);       // method 0
    h.u16(0);       // time
    h.u16(0x21);    // date 1980-01-01
    h.u32(r.crc);
    h.u32(r.size);
    h.u32(r.size);
    h.u16((uint32_t)name.size());
    h.u16(0);
    h.str(name);
    FILE* fp = static_cast<FILE*>(fp_);
    if (fwrite(h.b.data(), 1, h.b.size(), fp) != h.b.size() || (size && fwrite(data, 1, size, fp) != size)) {
        failed_ = true;
        return -1;
    }
    cursor_ += h.b.size() + size;
    records_.push_back(r);
    return 0;
}

int StoreZipWriter::close() {
    if (!fp_) return -1;
    FILE* fp = static_cast<FILE*>(fp_);
    ByteSink d;
    for (const Record& r : records_) {  // central directory
        d.u32(0x02014b50u);
        d.u16(10);  // made by
        d.u16(10);  // needed
        d.u16(0);
        d.u16(0);
        d.u16(0);
        d.u16(0x21);
        d.u32(r.crc);
        d.u32(r.size);
        d.u32(r.size);
        d.u16((uint32_t)r.name.size());
        d.u16(0);  // extra
        d.u16(0);  // comment
        d.u16(0);  // disk
        d.u16(0);  // internal attributes
        d.u32(0);  // external attributes
        d.u32(r.offset);
        d.str(r.name);
    }
    const size_t dir_bytes = d.b.size();
    bool ok = !failed_ && records_.size() <= 0xFFFFu && cursor_ + dir_bytes + 22 <= 0xFFFFFFFFull;
    d.u32(0x06054b50u);
    d.u16(0);
    d.u16(0);
    d.u16((uint32_t)records_.size());
    d.u16((uint32_t)records_.size());
    d.u32((uint32_t)dir_bytes);
    d.u32((uint32_t)cursor_);
    d.u16(0);
    ok = ok && fwrite(d.b.data(), 1, d.b.size(), fp) == d.b.size();
    ok = (fclose(fp) == 0) && ok;
    fp_ = nullptr;
    records_.clear();
    return ok ? 0 : -1;
}

}  // namespace pnnx
