// pnnx/storezip.h -- reader for the stored-only (method 0) ZIP container of a .pnnx.bin.
// Behaviour contract: reference src/pnnx/storezip.cpp:117-229 (sequential local-file-header scan,
// data-descriptor flag and compression rejected).  Own implementation: the archive is read into
// memory once and entries are slices of that buffer.
#ifndef SIMPLEINFER_AMD_PNNX_STOREZIP_H_
#define SIMPLEINFER_AMD_PNNX_STOREZIP_H_

#include <cstddef>
#include <map>
#include <string>
#include <vector>

namespace pnnx {

class StoreZipReader {
public:
    int open(const std::string& path);
    // 0 when the entry does not exist (as the reference's get_file_size does)
    size_t get_file_size(const std::string& name) const;
    int read_file(const std::string& name, char* data) const;
    void close();

private:
    struct Entry {
        size_t offset = 0;
        size_t size = 0;
    };
    std::vector<unsigned char> blob_;
    std::map<std::string, Entry> entries_;
};

}  // namespace pnnx

#endif  // SIMPLEINFER_AMD_PNNX_STOREZIP_H_
