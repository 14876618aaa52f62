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

}  // namespace pnnx
