// pnnx/storezip.h -- reader and writer for the stored-only (method 0) ZIP container of a .pnnx.bin.
// Behaviour contract: reference src/pnnx/storezip.cpp:117-229 (reader: sequential local-file-header scan,
// data-descriptor flag and compression rejected) and :242-395 (writer: stored entries + central directory,
// readable by any unzip).  Own implementation: the reader slurps the archive once and entries are slices of that
// buffer; the writer streams entries out and keeps only the directory records.
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

class StoreZipWriter {
public:
    ~StoreZipWriter();
    int open(const std::string& path);
    // one stored entry; sizes are limited to the classic (non-zip64) 4 GiB
    int write_file(const std::string& name, const char* data, size_t size);
    // writes the central directory; 0 when every byte reached the file
    int close();

private:
    struct Record {
        std::string name;
        unsigned int crc = 0;
        unsigned int size = 0;
        unsigned int offset = 0;
    };
    void* fp_ = nullptr;
    std::vector<Record> records_;
    unsigned long long cursor_ = 0;
    bool failed_ = false;
};

}  // namespace pnnx

#endif  // SIMPLEINFER_AMD_PNNX_STOREZIP_H_
