/*
 * include/cobs_gpu_construct.h -- C ABI of the construction side of libcobs_gpu.so (the step in
 * front of the query path): cobs::classic_construct / compact_construct / classic_combine /
 * classic_construct_random of the reference with term hashing and bit setting on the GPU, and the
 * cobs::DocumentList readers that feed them.  The query path -- the drop-in boundary of this library -- is
 * include/cobs_gpu.h; nothing there depends on this header.
 */
#ifndef COBS_GPU_CONSTRUCT_H
#define COBS_GPU_CONSTRUCT_H

#include "cobs_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- construction (the step in front of the query path) ------ */
typedef struct cobs_gpu_build_params {
    uint32_t struct_size;        /* sizeof(cobs_gpu_build_params) */
    uint32_t term_size;          /* ClassicIndexParameters::term_size, default 31 */
    uint32_t canonicalize;       /* 1 */
    uint32_t num_hashes;         /* 1 */
    double false_positive_rate;  /* 0.3 */
    uint64_t signature_size;     /* 0 = calc_signature_size(largest document, num_hashes, fpr) */
    uint64_t page_size;          /* compact: 0 = reference heuristic (compact_index.cpp:184-189) */
    int32_t device;              /* -1 = current */
    uint32_t text_batch_bytes;   /* documents are uploaded and hashed in batches of at most this much text
                                    (0 = 256 MiB): the batching of classic_index.cpp:565-659 without the
                                    per-batch index files -- every batch sets its bits in the one matrix in HBM */
    /* optional, ndocs entries: the number of terms of every document as the reference's document
     * index reports it (FastaFile::num_terms, fasta_file.hpp:147-153) -- what sizes a signature when
     * signature_size is 0.  NULL = count the k-grams of the given text. */
    const uint64_t* doc_terms;
    /* how build_kernel sets bits: 0 = automatic, 1 = atomicOr into the matrix, 2 = byte stores into
     * per-document planes that a second kernel packs into the matrix (faster than the part's
     * scattered-atomic rate; used when the planes of a batch fit 3 GiB).  Same index either way. */
    uint32_t set_bits_mode;
    uint32_t reserved;
} cobs_gpu_build_params;

/* classic_construct (construction/classic_index.cpp:565-659) for documents that are already
 * parsed: texts[d] holds the sequences of document d joined by '\n' (terms do not span a
 * separator), names[d] its name; documents are written in the given order.  Term hashing and
 * bit setting run on the GPU; the file is byte-for-byte what the reference writes. */
cobs_gpu_status cobs_gpu_build_classic(const char* const* names, const char* const* texts,
                                       const size_t* lens, size_t ndocs,
                                       const cobs_gpu_build_params* params, const char* out_path);
/* compact_construct (construction/compact_index.cpp:171-340): the documents must already be in
 * their final order (sorted by size, by path inside every group of 8*page_size); each group
 * becomes one sub-index with its own signature size. */
cobs_gpu_status cobs_gpu_build_compact(const char* const* names, const char* const* texts,
                                       const size_t* lens, size_t ndocs,
                                       const cobs_gpu_build_params* params, const char* out_path);

/* classic_construct (kind 0) / compact_construct (kind 1) straight into a resident query handle:
 * the bit matrix is built inside the handle's HBM blob at the engine's row pitch; no file, no host
 * copy (the reference always goes through a file: classic_index.cpp:565-659).  Arguments as
 * cobs_gpu_build_classic / _compact; opts may select the device (no shard, no budget). */
cobs_gpu_status cobs_gpu_build_index(uint32_t kind, const char* const* names, const char* const* texts,
                                     const size_t* lens, size_t ndocs, const cobs_gpu_build_params* params,
                                     const cobs_gpu_options* opts, cobs_gpu_index** out);
/* ---- documents: cobs::DocumentList / DocumentEntry and the reference's file readers -----------
 * (cobs/document_list.hpp:62-411; text_file.hpp, cortex_file.hpp, kmer_buffer.hpp, fasta_file.hpp,
 * fastq_file.hpp, fasta_multifile.hpp).  The list and the parsers live in the library (host
 * threads); the terms of the documents are hashed on the GPU by the *_list builders below. */
enum {
    COBS_GPU_FILETYPE_ANY = 0,          /* accept every supported type in a directory scan */
    COBS_GPU_FILETYPE_TEXT = 1,         /* .txt: every k-gram of the byte stream */
    COBS_GPU_FILETYPE_CORTEX = 2,       /* .ctx .cortex: McCortex v6, one colour */
    COBS_GPU_FILETYPE_KMER_BUFFER = 3,  /* .cobs_doc: packed 31-mers */
    COBS_GPU_FILETYPE_FASTA = 4,        /* .fa .fasta .fna .ffn .faa .frn (+ .gz): one document per file */
    COBS_GPU_FILETYPE_FASTQ = 5,        /* .fq .fastq (+ .gz): one document per file */
    COBS_GPU_FILETYPE_FASTA_MULTI = 6,  /* .mfasta: one document per '>' record */
    COBS_GPU_FILETYPE_FASTQ_MULTI = 7,  /* .mfastq: recognised, not loadable (as in the reference) */
    COBS_GPU_FILETYPE_LIST = 8,         /* .list: one path per line */
    COBS_GPU_FILETYPE_MEMORY = 9        /* a document handed over with cobs_gpu_doclist_add_memory */
};
enum { COBS_GPU_SORT_BY_PATH = 0, COBS_GPU_SORT_BY_SIZE = 1 };

typedef struct cobs_gpu_doclist cobs_gpu_doclist;
typedef struct cobs_gpu_doc_entry {     /* cobs::DocumentEntry, document_list.hpp:62-76 */
    const char* path;                   /* owned by the list, valid until it changes */
    const char* name;
    uint32_t type;                      /* COBS_GPU_FILETYPE_* */
    uint32_t reserved;
    uint64_t size;                      /* size_: bytes, or characters of a sub-document */
    uint64_t subdoc_index;
    uint64_t term_size;                 /* fixed k-mer size of the file, or 0 */
    uint64_t term_count;
} cobs_gpu_doc_entry;

cobs_gpu_status cobs_gpu_doclist_create(cobs_gpu_doclist** out);
void cobs_gpu_doclist_free(cobs_gpu_doclist* dl);
/* DocumentList::add (document_list.hpp:337-340): identify the file by its extension and append its
 * document(s) */
cobs_gpu_status cobs_gpu_doclist_add(cobs_gpu_doclist* dl, const char* path);
/* DocumentList::add_recursive (:345-411): a directory is scanned recursively for files of the
 * filter type, a .list file is read, a single file is added; the list is then sorted by path */
cobs_gpu_status cobs_gpu_doclist_add_recursive(cobs_gpu_doclist* dl, const char* root, uint32_t filter);
/* an in-memory document: its sequences joined by '\n' (no reference counterpart; the texts of
 * cobs_gpu_build_classic as list entries) */
cobs_gpu_status cobs_gpu_doclist_add_memory(cobs_gpu_doclist* dl, const char* name, const char* text, size_t len);
size_t cobs_gpu_doclist_size(const cobs_gpu_doclist* dl);
cobs_gpu_status cobs_gpu_doclist_entry(const cobs_gpu_doclist* dl, size_t i, cobs_gpu_doc_entry* out);
cobs_gpu_status cobs_gpu_doclist_sort(cobs_gpu_doclist* dl, uint32_t by);           /* sort_by_path / sort_by_size */
/* DocumentEntry::num_terms(k) (:85-112): the count that sizes a signature */
cobs_gpu_status cobs_gpu_doclist_num_terms(const cobs_gpu_doclist* dl, size_t i, uint32_t term_size, uint64_t* out);
/* DocumentEntry::process_terms(k, callback) (:116-151) on the host, for `cobs doc-dump` /
 * `print-kmers` style callers: the document's terms back to back, term_size bytes each, as many
 * as fit cap_bytes; *n_terms receives how many there are (call with cap_bytes 0 to size). */
cobs_gpu_status cobs_gpu_doclist_terms(const cobs_gpu_doclist* dl, size_t i, uint32_t term_size,
                                       char* out, size_t cap_bytes, uint64_t* n_terms);
/* StringToFileType (cobs/document_list.cpp:15-32): "any" "text" "cortex" "cobs_doc" "fasta" "fastq" "list" */
cobs_gpu_status cobs_gpu_filetype_from_string(const char* s, uint32_t* out);

/* classic_construct (classic_index.cpp:565-659) from a document list: documents in list order, one
 * signature size from the num_terms of the largest document by (size, path) (:521-563); the files
 * are read and parsed by host threads while the GPU hashes the previous batch. */
cobs_gpu_status cobs_gpu_build_classic_list(const cobs_gpu_doclist* dl, const cobs_gpu_build_params* params,
                                            const char* out_path);
/* compact_construct (compact_index.cpp:171-340) from a document list: sorted by (size, path), cut
 * into groups of 8 * page_size documents, every group in (path, sub-document) order with the
 * signature size of its largest num_terms; a group without terms is left out (:285-286). */
cobs_gpu_status cobs_gpu_build_compact_list(const cobs_gpu_doclist* dl, const cobs_gpu_build_params* params,
                                            const char* out_path);
/* the same straight into a resident query handle (cf. cobs_gpu_build_index) */
cobs_gpu_status cobs_gpu_build_index_list(uint32_t kind, const cobs_gpu_doclist* dl, const cobs_gpu_build_params* params,
                                          const cobs_gpu_options* opts, cobs_gpu_index** out);

/* The builders keep their staging memory (three pinned + device text buffers of 256 MiB, the byte
 * planes) for the next build of the process; this frees what no build is using right now. */
void cobs_gpu_build_release_buffers(void);

/* classic_combine (construction/classic_index.cpp:195-327): the rows of n classic indexes with equal
 * term size / canonicalize / hashes / signature size concatenated at bit granularity into one
 * index, document names in input order; row batches of at most mem_bytes (0 = 1 GiB) are
 * interleaved on the device. */
cobs_gpu_status cobs_gpu_combine_classic(const char* const* in_paths, size_t n, const char* out_path,
                                         uint64_t mem_bytes, int device);
/* compact_combine_into_compact (compact_index.cpp:51-169; `cobs compact-construct-combine`): n
 * classic indexes with equal term size / canonicalize become the sub-indexes of one compact index
 * (own signature size and hash count each), rows padded to page_size bytes; every input but the
 * last must have a row size of exactly page_size.  File to file, no device work; unlike the
 * reference the inputs are not deleted. */
cobs_gpu_status cobs_gpu_combine_compact(const char* const* in_paths, size_t n, const char* out_path, uint64_t page_size);
/* classic_construct_random (classic_index.cpp:661-725; `cobs classic-construct-random`,
 * src/cobs.cpp:243-291): num_documents documents of document_size random 31-mers, canonicalised,
 * hashed num_hashes times into signature_size rows, written as a .cobs_classic file.  Same
 * distribution as the reference, not the same random stream. */
cobs_gpu_status cobs_gpu_construct_random(const char* out_path, uint64_t signature_size, uint64_t num_documents,
                                          uint64_t document_size, uint64_t num_hashes, uint64_t seed, int device);

#ifdef __cplusplus
}
#endif
#endif /* COBS_GPU_CONSTRUCT_H */
