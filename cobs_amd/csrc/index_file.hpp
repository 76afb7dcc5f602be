// cobs_amd/csrc/index_file.hpp -- read side of the two COBS index file formats.
//
// Byte layout (little-endian PODs written raw by the reference):
//   classic  reference cobs/file/classic_index_header.cpp:26-50
//   compact  reference cobs/file/compact_index_header.cpp:20-65
//   magic    reference cobs/file/header.hpp:22-59
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace cobs_amd {

enum class IndexKind : uint32_t { Classic = 0, Compact = 1 };

// Geometry of one index file: what the reference exposes through
// IndexSearchFile (cobs/query/index_file.hpp:19-35).
struct IndexMeta {
    IndexKind kind = IndexKind::Classic;
    uint32_t term_size = 0;
    uint8_t canonicalize = 0;
    uint64_t num_hashes = 0;
    uint64_t header_page_size = 0;              // compact: bytes per sub-index row; classic: unused
    std::vector<uint64_t> signature_sizes;      // rows per sub-index (classic: one entry)
    std::vector<std::string> doc_names;         // file_names()
    uint64_t data_offset = 0;                   // first matrix byte in the file

    uint32_t num_pages() const { return (uint32_t)signature_sizes.size(); }
    // bytes per row of one sub-index as stored in the file
    uint64_t page_row_bytes() const {
        return kind == IndexKind::Classic ? (doc_names.size() + 7) / 8 : header_page_size;
    }
    // IndexSearchFile::row_size(): classic ceil(D/8); compact page_size * P
    uint64_t row_size() const { return page_row_bytes() * num_pages(); }
    // IndexSearchFile::counts_size(): score slots incl. padding documents
    uint64_t counts_size() const { return 8 * row_size(); }
    // IndexSearchFile::page_size(): the classic search file reports 1
    uint64_t page_size() const { return kind == IndexKind::Classic ? 1 : header_page_size; }
    // file offset of row 0 of sub-index p
    uint64_t page_offset(uint32_t p) const {
        uint64_t off = data_offset;
        for (uint32_t i = 0; i < p; ++i) off += page_row_bytes() * signature_sizes[i];
        return off;
    }
    uint64_t data_bytes() const {
        uint64_t n = 0;
        for (uint64_t s : signature_sizes) n += page_row_bytes() * s;
        return n;
    }
};

// Parse the header found at data[0..len).  Tries the classic layout first and
// the compact layout second, like ClassicSearch(std::string path) does
// (reference cobs/query/classic_search.cpp:51-64).  Returns false and sets err
// if neither matches.
bool parse_index_header(const uint8_t* data, size_t len, IndexMeta& meta, std::string& err);

// A read-only memory map of a whole file.
class MappedFile {
public:
    MappedFile() = default;
    ~MappedFile();
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    bool open(const std::string& path, std::string& err);
    const uint8_t* data() const { return data_; }
    size_t size() const { return size_; }

private:
    const uint8_t* data_ = nullptr;
    size_t size_ = 0;
    int fd_ = -1;
};

}  // namespace cobs_amd
