// cobs_amd/csrc/documents.hpp -- the documents in front of index construction: what the
// reference's DocumentList / DocumentEntry and its file readers (cobs/document_list.hpp,
// text_file.hpp, cortex_file.hpp, kmer_buffer.hpp, fasta_file.hpp, fastq_file.hpp,
// fasta_multifile.hpp) hand to classic_construct / compact_construct, restated as "term text":
// the bytes whose k-grams are the document's terms, ready for build_kernel.
#pragma once

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/cobs_gpu_construct.h"

namespace cobs_amd {

// cobs::FileType (document_list.hpp:35-54); Memory = a document handed over by the caller
enum class FileType : uint32_t {
    Any = COBS_GPU_FILETYPE_ANY, Text = COBS_GPU_FILETYPE_TEXT, Cortex = COBS_GPU_FILETYPE_CORTEX,
    KMerBuffer = COBS_GPU_FILETYPE_KMER_BUFFER, Fasta = COBS_GPU_FILETYPE_FASTA, Fastq = COBS_GPU_FILETYPE_FASTQ,
    FastaMulti = COBS_GPU_FILETYPE_FASTA_MULTI, FastqMulti = COBS_GPU_FILETYPE_FASTQ_MULTI,
    List = COBS_GPU_FILETYPE_LIST, Memory = COBS_GPU_FILETYPE_MEMORY,
};

// cobs::DocumentEntry (document_list.hpp:62-76) plus what the readers keep between the index
// pass and process_terms (the reference keeps the same in its .cobs_cache files)
struct DocEntry {
    std::string path, name;
    FileType type = FileType::Any;
    uint64_t size = 0, subdoc_index = 0, term_size = 0, term_count = 0;
    std::map<uint64_t, uint64_t> run_hist;   // Fasta / Fastq: sequence length -> how many
    uint64_t pos_begin = 0;                  // FastaMulti: file offset of the sub-document's first line
    uint64_t data_bytes = 0;                 // Fasta / Fastq: (inflated) bytes of the file when it was listed
    std::string text;                        // Memory: sequences joined by '\n'
};

// Where a reader writes term text: a caller-provided span (pinned staging memory during a build).
struct TermSink {
    char* data = nullptr;
    size_t cap = 0, size = 0;
    bool overflow = false;
    void put(const char* p, size_t n) {
        if (size + n > cap) { overflow = true; return; }
        std::memcpy(data + size, p, n);
        size += n;
    }
    void put(char c) { put(&c, 1); }
};

// One stretch of term text (offsets relative to the sink).  raw = every k-gram inside [begin, begin + len) is a term (newlines
// are ordinary characters); otherwise the stretch is sequences each FOLLOWED by '\n' and no term
// holds a '\n'.
struct TermSeg {
    uint64_t begin, len;
    bool raw;
};

FileType identify_filetype(const std::string& path);                              // document_list.hpp:199-243
bool parse_filetype(const std::string& s, FileType& out);                         // StringToFileType
//! DocumentList::load: the entries one path contributes (several for a multi-FASTA file)
cobs_gpu_status load_entries(const std::string& path, std::vector<DocEntry>& out);
//! DocumentList::add_recursive: directory scan / .list file / single file, then sorted
cobs_gpu_status add_recursive(const std::string& root, FileType filter, std::vector<DocEntry>& list);
//! by = COBS_GPU_SORT_BY_PATH: (path, sub-document) -- DocumentEntry::operator<; _BY_SIZE: (size, path)
void sort_entries(std::vector<DocEntry>& list, uint32_t by);
//! DocumentEntry::num_terms(k) -- the count that sizes a signature
uint64_t num_terms(const DocEntry& e, uint32_t k);
//! an upper bound of the term text load_terms writes for this entry
uint64_t term_text_bound(const DocEntry& e, uint32_t k);
//! DocumentEntry::process_terms(k) as term text written to `out`; `scratch` holds the file while it
//! is parsed (reused by the caller from document to document)
cobs_gpu_status load_terms(const DocEntry& e, uint32_t k, TermSink& out, std::vector<TermSeg>& segs, std::string& scratch);

}  // namespace cobs_amd

// the handle of the C ABI
struct cobs_gpu_doclist {
    std::vector<cobs_amd::DocEntry> list;
};
