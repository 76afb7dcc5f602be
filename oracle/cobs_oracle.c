/*
 * oracle/cobs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the bingmann/cobs query path, used only as the
 * parity checker (tests/, __graft_entry__.smoke()) and as the timed CPU baseline
 * ("port") of bench.py.  See cobs_oracle.h for the pinning status.
 *
 * Every function names the reference location it follows (paths are relative
 * to the reference checkout, e.g. cobs/query/classic_search.cpp:403-505).
 */
#define _GNU_SOURCE
#include "cobs_oracle.h"

#include <emmintrin.h>
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------------ */
/* errors                                                                    */

static __thread char g_err[512];

static int fail(int code, const char* fmt, const char* arg) {
    snprintf(g_err, sizeof g_err, fmt, arg ? arg : "");
    return code;
}

const char* oracle_last_error(void) { return g_err; }

/* ------------------------------------------------------------------------ */
/* XXH64 -- third-party xxHash (reference submodule extlib/xxhash, not vendored),
 * restated from the public XXH64 specification.  Reference call sites:
 * cobs/query/classic_search.cpp:84,99 and cobs/util/misc.hpp:69.              */

#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) {
    return rotl64(acc + in * XP2, 31) * XP1;
}
static inline uint64_t xmerge(uint64_t h, uint64_t v) {
    return (h ^ xround(0, v)) * XP1 + XP4;
}

uint64_t oracle_xxh64(const void* data, size_t len, uint64_t seed) {
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t* lim = end - 32;
        do {
            v1 = xround(v1, rd64(p));
            v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16));
            v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) {
        h ^= xround(0, rd64(p));
        h = rotl64(h, 27) * XP1 + XP4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= (uint64_t)rd32(p) * XP1;
        h = rotl64(h, 23) * XP2 + XP3;
        p += 4;
    }
    while (p < end) {
        h ^= (uint64_t)(*p) * XP5;
        h = rotl64(h, 11) * XP1;
        p++;
    }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------------ */
/* canonicalize_kmer -- cobs/util/query.cpp:143-199 (maps :104-141).
 * Scan the first k/2 positions comparing the forward base with the complement
 * of the mirrored base; first strict inequality decides; the middle base of an
 * odd k is never compared; ties keep the forward k-mer.                        */

static inline char fwd_base(uint8_t c) {
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? (char)c : 0;
}
static inline char rev_base(uint8_t c) {
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 0;
    }
}

int oracle_canonicalize_kmer(const char* in, char* out, size_t k) {
    const uint8_t* s = (const uint8_t*)in;
    int good = 1;
    size_t i = 0;
    for (; i < k / 2; ++i) {
        char f = fwd_base(s[i]);
        char r = rev_base(s[k - 1 - i]);
        out[i] = f;
        good = good && f != 0 && r != 0;
        if (f < r) {
            for (++i; i < k; ++i) {
                char g = fwd_base(s[i]);
                out[i] = g;
                good = good && g != 0;
            }
            return good;
        }
        if (f > r) {
            for (size_t j = 0; j < k; ++j) {
                char x = rev_base(s[j]);
                out[k - 1 - j] = x;
                good = good && x != 0;
            }
            return good;
        }
    }
    for (; i < k; ++i) {
        char g = fwd_base(s[i]);
        out[i] = g;
        good = good && g != 0;
    }
    return good;
}

/* term hashing shared by the query path (classic_search.cpp:66-107) and the
 * construction side (cobs/util/misc.hpp:65-72, construction/classic_index.cpp:46-73) */
void oracle_term_hashes(const char* seq, size_t len, uint32_t k, int canonicalize,
                        uint64_t num_hashes, uint64_t* out, uint8_t* good) {
    if (len < k) return;
    char* buf = (char*)malloc(k ? k : 1);
    for (size_t i = 0; i + k <= len; ++i) {
        const char* term = seq + i;
        int g = 1;
        if (canonicalize) { g = oracle_canonicalize_kmer(seq + i, buf, k); term = buf; }
        if (good) good[i] = (uint8_t)g;
        for (uint64_t j = 0; j < num_hashes; ++j) out[i * num_hashes + j] = oracle_xxh64(term, k, j);
    }
    free(buf);
}

/* ------------------------------------------------------------------------ */
/* query generators of the reference's tests and benchmark                   */

/* cobs/util/misc.cpp:32-35 + misc.hpp:30-38: std::default_random_engine is
 * minstd_rand0 (x <- 16807 x mod 2^31-1; seed 0 is mapped to 1).              */
void oracle_random_sequence(char* out, size_t size, uint64_t seed) {
    static const char bp[4] = { 'A', 'C', 'G', 'T' };
    uint64_t x = seed % 2147483647ULL;
    if (x == 0) x = 1;
    for (size_t i = 0; i < size; ++i) {
        x = (x * 16807ULL) % 2147483647ULL;
        out[i] = bp[x % 4];
    }
}

/* src/cobs.cpp:709-720: std::mt19937 rng(seed); rng() % 4 per character, one
 * generator shared by all warm-up and benchmark queries.                     */
void oracle_mt19937_sequence(char* out, size_t size, uint32_t* st, uint32_t seed,
                             int first) {
    static const char bp[4] = { 'A', 'C', 'G', 'T' };
    uint32_t* mt = st;
    uint32_t* idx = st + 624;
    if (first) {
        mt[0] = seed;
        for (uint32_t i = 1; i < 624; ++i)
            mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
        *idx = 624;
    }
    for (size_t c = 0; c < size; ++c) {
        if (*idx >= 624) {
            for (uint32_t i = 0; i < 624; ++i) {
                uint32_t y = (mt[i] & 0x80000000U) | (mt[(i + 1) % 624] & 0x7fffffffU);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
            }
            *idx = 0;
        }
        uint32_t y = mt[(*idx)++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680U;
        y ^= (y << 15) & 0xefc60000U;
        y ^= y >> 18;
        out[c] = bp[y % 4];
    }
}

/* ------------------------------------------------------------------------ */
/* procedural synthetic index bits (shared definition with the HIP library's
 * synthetic generator; NOT part of the reference).  Bit density ~0.297.       */

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint64_t synth_word(uint64_t seed, uint32_t page, uint64_t row, uint64_t w) {
    uint64_t key = mix64(seed ^ mix64(((uint64_t)page << 40) ^ row));
    uint64_t c = key + w * 6;
    uint64_t a = mix64(c) & mix64(c + 1);
    uint64_t b = mix64(c + 2) & mix64(c + 3) & mix64(c + 4) & mix64(c + 5);
    return a | b;
}

void oracle_synth_fill(int kind, uint64_t seed, uint64_t page_size, uint32_t num_pages,
                       uint32_t num_docs, uint32_t page, uint64_t row,
                       uint64_t byte_begin, uint64_t n, uint8_t* out) {
    /* bytes per row of this sub-index, and how many of its documents are real */
    uint64_t rsz = kind == 0 ? ((uint64_t)num_docs + 7) / 8 : page_size;
    uint64_t first_doc = kind == 0 ? 0 : (uint64_t)page * 8 * page_size;
    uint64_t live = num_docs > first_doc ? num_docs - first_doc : 0;   /* docs with bits */
    (void)num_pages;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t b = byte_begin + i;
        uint8_t v = 0;
        if (b < rsz) {
            v = (uint8_t)(synth_word(seed, page, row, b >> 3) >> (8 * (b & 7)));
            if (b * 8 >= live) v = 0;
            else if (b * 8 + 8 > live) v &= (uint8_t)((1u << (live - b * 8)) - 1u);
        }
        out[i] = v;
    }
}

/* ------------------------------------------------------------------------ */
/* index files -- cobs/file/classic_index_header.cpp:26-50,
 * cobs/file/compact_index_header.cpp:20-65, cobs/file/header.hpp:22-59        */

struct oracle_index {
    int kind;                 /* 0 classic, 1 compact */
    uint32_t term_size;
    uint8_t canonicalize;
    uint64_t num_hashes;
    uint64_t page_size;       /* compact: header page_size; classic: row_size (internal) */
    uint32_t num_pages;
    uint64_t* sig;            /* signature_size per sub-index */
    uint32_t num_docs;
    char** names;
    uint64_t data_off;
    /* storage */
    uint8_t* map; size_t map_len; int fd;
    const uint8_t** page_data;
    int synthetic; uint64_t seed;
    /* planted true positives of a procedural index (oracle_plant): (page << 40 | row, bit of the row) pairs, sorted
     * by key before the first lookup */
    uint64_t* plant_key; uint32_t* plant_bit; size_t plant_n, plant_cap; int plant_sorted;
};

static const char MAGIC[] = "COBS:";
static const char CLASSIC_WORD[] = "CLASSIC_INDEX";
static const char COMPACT_WORD[] = "COMPACT_INDEX";

typedef struct { const uint8_t* p; size_t len; size_t pos; int bad; } reader;

static void rd_bytes(reader* r, void* dst, size_t n) {
    if (r->bad || r->pos + n > r->len) { r->bad = 1; memset(dst, 0, n); return; }
    memcpy(dst, r->p + r->pos, n); r->pos += n;
}
static int rd_word(reader* r, const char* w) {
    size_t n = strlen(w);
    if (r->bad || r->pos + n > r->len || memcmp(r->p + r->pos, w, n) != 0) { r->bad = 1; return 0; }
    r->pos += n; return 1;
}
/* std::getline: up to and excluding '\n' */
static char* rd_line(reader* r) {
    if (r->bad) return NULL;
    size_t s = r->pos;
    while (r->pos < r->len && r->p[r->pos] != '\n') r->pos++;
    if (r->pos >= r->len) { r->bad = 1; return NULL; }
    size_t n = r->pos - s;
    char* out = (char*)malloc(n + 1);
    memcpy(out, r->p + s, n); out[n] = 0;
    r->pos++;
    return out;
}

static uint64_t ix_row_size(const oracle_index* ix) {
    /* classic_index_header.cpp:22-24 ; compact_index/search_file.cpp:21 */
    return ix->kind == 0 ? ((uint64_t)ix->num_docs + 7) / 8 : ix->page_size * ix->num_pages;
}

static void free_index(oracle_index* ix) {
    if (!ix) return;
    if (ix->names) { for (uint32_t i = 0; i < ix->num_docs; ++i) free(ix->names[i]); free(ix->names); }
    free(ix->sig); free((void*)ix->page_data);
    free(ix->plant_key); free(ix->plant_bit);
    if (ix->map) munmap(ix->map, ix->map_len);
    if (ix->fd >= 0) close(ix->fd);
    free(ix);
}

static int parse_classic(reader* r, oracle_index* ix) {
    uint32_t version = 0, nfiles = 0; uint64_t sig = 0;
    if (!rd_word(r, MAGIC) || !rd_word(r, CLASSIC_WORD)) return 0;
    rd_bytes(r, &version, 4);
    if (r->bad || version != 1) return 0;
    rd_bytes(r, &ix->term_size, 4); rd_bytes(r, &ix->canonicalize, 1);
    rd_bytes(r, &nfiles, 4); rd_bytes(r, &sig, 8); rd_bytes(r, &ix->num_hashes, 8);
    if (r->bad) return 0;
    ix->kind = 0; ix->num_docs = nfiles; ix->num_pages = 1;
    ix->sig = (uint64_t*)malloc(8); ix->sig[0] = sig;
    ix->names = (char**)calloc(nfiles ? nfiles : 1, sizeof(char*));
    for (uint32_t i = 0; i < nfiles; ++i) { ix->names[i] = rd_line(r); if (!ix->names[i]) return 0; }
    if (!rd_word(r, CLASSIC_WORD)) return 0;
    ix->data_off = r->pos;
    ix->page_size = ((uint64_t)nfiles + 7) / 8;
    return 1;
}

static int parse_compact(reader* r, oracle_index* ix) {
    uint32_t version = 0, nparams = 0, nfiles = 0;
    if (!rd_word(r, MAGIC) || !rd_word(r, COMPACT_WORD)) return 0;
    rd_bytes(r, &version, 4);
    if (r->bad || version != 1) return 0;
    rd_bytes(r, &ix->term_size, 4); rd_bytes(r, &ix->canonicalize, 1);
    rd_bytes(r, &nparams, 4); rd_bytes(r, &nfiles, 4); rd_bytes(r, &ix->page_size, 8);
    if (r->bad || nparams == 0 || ix->page_size == 0) return 0;
    ix->kind = 1; ix->num_docs = nfiles; ix->num_pages = nparams;
    ix->sig = (uint64_t*)malloc(8 * (size_t)nparams);
    for (uint32_t p = 0; p < nparams; ++p) {
        uint64_t nh = 0;
        rd_bytes(r, &ix->sig[p], 8); rd_bytes(r, &nh, 8);
        if (p == 0) ix->num_hashes = nh;
        else if (nh != ix->num_hashes) return 0;     /* compact_index/search_file.cpp:24-27 */
    }
    ix->names = (char**)calloc(nfiles ? nfiles : 1, sizeof(char*));
    for (uint32_t i = 0; i < nfiles; ++i) { ix->names[i] = rd_line(r); if (!ix->names[i]) return 0; }
    /* compact_index_header.cpp:20-22: pad so that the data after the closing
     * magic word starts at a multiple of page_size */
    uint64_t pad = (ix->page_size - ((r->pos + strlen(COMPACT_WORD)) % ix->page_size)) % ix->page_size;
    r->pos += pad;
    if (!rd_word(r, COMPACT_WORD)) return 0;
    ix->data_off = r->pos;
    return 1;
}

int oracle_open(const char* path, oracle_index** out) {
    *out = NULL;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ORACLE_ERR_OPEN, "could not open index file %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) {
        close(fd); return fail(ORACLE_ERR_OPEN, "not a regular file: %s", path);
    }
    uint8_t* m = (uint8_t*)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); return fail(ORACLE_ERR_OPEN, "mmap failed: %s", path); }
    oracle_index* ix = (oracle_index*)calloc(1, sizeof *ix);
    ix->fd = fd; ix->map = m; ix->map_len = (size_t)st.st_size;
    /* format sniffing: try classic, then compact (classic_search.cpp:51-64) */
    reader r = { m, (size_t)st.st_size, 0, 0 };
    int ok = parse_classic(&r, ix);
    if (!ok) {
        if (ix->names) { for (uint32_t i = 0; i < ix->num_docs; ++i) free(ix->names[i]); free(ix->names); ix->names = NULL; }
        free(ix->sig); ix->sig = NULL;
        reader r2 = { m, (size_t)st.st_size, 0, 0 };
        ok = parse_compact(&r2, ix);
    }
    if (!ok) { free_index(ix); return fail(ORACLE_ERR_FORMAT, "Could not open index path \"%s\"", path); }
    /* sub-index data pointers: compact_index/mmap_search_file.cpp:17-28 */
    ix->page_data = (const uint8_t**)malloc(sizeof(uint8_t*) * ix->num_pages);
    uint64_t off = ix->data_off;
    for (uint32_t p = 0; p < ix->num_pages; ++p) {
        ix->page_data[p] = m + off;
        off += ix->page_size * ix->sig[p];
    }
    if (off > (uint64_t)st.st_size) { free_index(ix); return fail(ORACLE_ERR_FORMAT, "index file truncated: %s", path); }
    *out = ix;
    return ORACLE_OK;
}

static oracle_index* make_mem(int kind, uint32_t term_size, uint8_t canonicalize,
                              uint64_t num_hashes, uint64_t page_size, uint32_t num_pages,
                              const uint64_t* sigs, uint32_t num_docs) {
    oracle_index* ix = (oracle_index*)calloc(1, sizeof *ix);
    ix->fd = -1; ix->kind = kind; ix->term_size = term_size; ix->canonicalize = canonicalize;
    ix->num_hashes = num_hashes; ix->num_pages = kind == 0 ? 1 : num_pages; ix->num_docs = num_docs;
    ix->page_size = kind == 0 ? ((uint64_t)num_docs + 7) / 8 : page_size;
    ix->sig = (uint64_t*)malloc(8 * (size_t)ix->num_pages);
    memcpy(ix->sig, sigs, 8 * (size_t)ix->num_pages);
    ix->names = (char**)calloc(num_docs ? num_docs : 1, sizeof(char*));
    for (uint32_t i = 0; i < num_docs; ++i) {
        char buf[32]; snprintf(buf, sizeof buf, "file_%06u", i);   /* classic_index.cpp:668-670 */
        ix->names[i] = strdup(buf);
    }
    ix->page_data = (const uint8_t**)calloc(ix->num_pages, sizeof(uint8_t*));
    return ix;
}

int oracle_from_memory(int kind, uint32_t term_size, uint8_t canonicalize,
                       uint64_t num_hashes, uint64_t page_size, uint32_t num_pages,
                       const uint64_t* sigs, uint32_t num_docs,
                       const uint8_t* const* page_data, oracle_index** out) {
    oracle_index* ix = make_mem(kind, term_size, canonicalize, num_hashes, page_size, num_pages, sigs, num_docs);
    for (uint32_t p = 0; p < ix->num_pages; ++p) ix->page_data[p] = page_data[p];
    *out = ix;
    return ORACLE_OK;
}

int oracle_synthetic(int kind, uint32_t term_size, uint8_t canonicalize,
                     uint64_t num_hashes, uint64_t page_size, uint32_t num_pages,
                     const uint64_t* sigs, uint32_t num_docs, uint64_t seed, oracle_index** out) {
    oracle_index* ix = make_mem(kind, term_size, canonicalize, num_hashes, page_size, num_pages, sigs, num_docs);
    ix->synthetic = 1; ix->seed = seed;
    *out = ix;
    return ORACLE_OK;
}

void oracle_close(oracle_index* ix) { free_index(ix); }

uint32_t oracle_term_size(const oracle_index* ix) { return ix->term_size; }
uint32_t oracle_canonicalize(const oracle_index* ix) { return ix->canonicalize; }
uint64_t oracle_num_hashes(const oracle_index* ix) { return ix->num_hashes; }
/* classic_index/search_file.hpp:26 page_size()==1 */
uint64_t oracle_page_size(const oracle_index* ix) { return ix->kind == 0 ? 1 : ix->page_size; }
uint64_t oracle_row_size(const oracle_index* ix) { return ix_row_size(ix); }
/* classic_index/search_file.cpp:21-23 ; compact_index/search_file.cpp:30-32 */
uint64_t oracle_counts_size(const oracle_index* ix) { return 8 * ix_row_size(ix); }
uint32_t oracle_num_pages(const oracle_index* ix) { return ix->num_pages; }
uint64_t oracle_signature_size(const oracle_index* ix, uint32_t p) { return ix->sig[p]; }
uint32_t oracle_num_docs(const oracle_index* ix) { return ix->num_docs; }
const char* oracle_doc_name(const oracle_index* ix, uint32_t d) { return d < ix->num_docs ? ix->names[d] : ""; }
uint64_t oracle_data_offset(const oracle_index* ix) { return ix->data_off; }

/* ------------------------------------------------------------------------ */
/* phase timers (cobs/util/timer.cpp; phases of classic_search.cpp:329-392)   */

static pthread_mutex_t g_tmutex = PTHREAD_MUTEX_INITIALIZER;
static double g_timers[5];
enum { T_HASH = 0, T_IO = 1, T_AND = 2, T_ADD = 3, T_SORT = 4 };

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void timer_add(const double t[5]) {
    pthread_mutex_lock(&g_tmutex);
    for (int i = 0; i < 5; ++i) g_timers[i] += t[i];
    pthread_mutex_unlock(&g_tmutex);
}
void oracle_timers(double out[5], int reset) {
    pthread_mutex_lock(&g_tmutex);
    if (out) memcpy(out, g_timers, sizeof g_timers);
    if (reset) memset(g_timers, 0, sizeof g_timers);
    pthread_mutex_unlock(&g_tmutex);
}

/* ------------------------------------------------------------------------ */
/* create_hashes -- classic_search.cpp:66-107                                */

static int create_hashes(const oracle_index* ix, const char* q, size_t qlen, uint64_t* hashes) {
    uint32_t k = ix->term_size;
    uint64_t H = ix->num_hashes;
    size_t T = qlen - k + 1;
    if (ix->canonicalize == 0) {
        for (size_t i = 0; i < T; ++i)
            for (uint64_t j = 0; j < H; ++j)
                hashes[i * H + j] = oracle_xxh64(q + i, k, j);
        return ORACLE_OK;
    }
    if (ix->canonicalize != 1) return fail(ORACLE_ERR_FORMAT, "Unknown canonicalize value%s", "");
    char* buf = (char*)malloc(k ? k : 1);
    for (size_t i = 0; i < T; ++i) {
        if (!oracle_canonicalize_kmer(q + i, buf, k)) {
            free(buf);
            return fail(ORACLE_ERR_INVALID_BASE, "Invalid DNA base pair in query string. Only ACGT are allowed.%s", "");
        }
        for (uint64_t j = 0; j < H; ++j) hashes[i * H + j] = oracle_xxh64(buf, k, j);
    }
    free(buf);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* True positives planted into a procedural index -- NOT part of the reference: the checker's restatement of the HIP
 * library's cobs_gpu_plant (include/cobs_gpu_batch.h).  Document docs[i] additionally holds term t of `text` iff
 * mix64(salt ^ (uint64_t)docs[i] << 32 ^ t) % 1000 < keep_permille[i]; a held term sets, for every hash function, the
 * document's bit in row hash % S_p of its sub-index -- what construction does for a document's own terms
 * (cobs/construction/classic_index.cpp:40-73: bit doc % 8 of byte doc / 8 of the row).  The hashes are the query
 * side's (create_hashes above), so a query that contains the term finds the bit.                                   */

static pthread_mutex_t g_plant_mu = PTHREAD_MUTEX_INITIALIZER;

int oracle_plant(oracle_index* ix, const char* text, size_t len, const uint32_t* docs,
                 const uint32_t* keep_permille, size_t ndocs, uint64_t salt) {
    if (!ix->synthetic) return ORACLE_ERR_ARG;
    uint32_t k = ix->term_size;
    if (len < k || ndocs == 0) return ORACLE_OK;
    size_t T = len - k + 1;
    uint64_t H = ix->num_hashes;
    uint64_t* hashes = (uint64_t*)malloc(T * H * sizeof(uint64_t));
    int rc = create_hashes(ix, text, len, hashes);
    if (rc != ORACLE_OK) { free(hashes); return rc; }
    uint64_t page_docs = ix->kind == 0 ? ~0ull : 8 * ix->page_size;
    pthread_mutex_lock(&g_plant_mu);
    for (size_t i = 0; i < ndocs; ++i) {
        uint64_t d = docs[i];
        if (d >= ix->num_docs || keep_permille[i] > 1000) { pthread_mutex_unlock(&g_plant_mu); free(hashes); return ORACLE_ERR_ARG; }
        uint32_t p = ix->kind == 0 ? 0 : (uint32_t)(d / page_docs);
        uint32_t bit = (uint32_t)(ix->kind == 0 ? d : d - (uint64_t)p * page_docs);
        for (size_t t = 0; t < T; ++t) {
            if (mix64(salt ^ (d << 32) ^ (uint64_t)t) % 1000u >= keep_permille[i]) continue;
            for (uint64_t j = 0; j < H; ++j) {
                if (ix->plant_n == ix->plant_cap) {
                    ix->plant_cap = ix->plant_cap ? 2 * ix->plant_cap : 1u << 16;
                    ix->plant_key = (uint64_t*)realloc(ix->plant_key, ix->plant_cap * sizeof(uint64_t));
                    ix->plant_bit = (uint32_t*)realloc(ix->plant_bit, ix->plant_cap * sizeof(uint32_t));
                }
                ix->plant_key[ix->plant_n] = ((uint64_t)p << 44) | (hashes[t * H + j] % ix->sig[p]);
                ix->plant_bit[ix->plant_n] = bit;
                ix->plant_n++;
            }
        }
    }
    ix->plant_sorted = 0;
    pthread_mutex_unlock(&g_plant_mu);
    free(hashes);
    return ORACLE_OK;
}

/* sort the planted entries by (page, row): an index sort, then both arrays permuted */
static const uint64_t* g_sort_keys;
static int cmp_plant(const void* a, const void* b) {
    uint64_t x = g_sort_keys[*(const uint32_t*)a], y = g_sort_keys[*(const uint32_t*)b];
    return x < y ? -1 : x > y;
}
static void plant_sort(oracle_index* ix) {
    pthread_mutex_lock(&g_plant_mu);
    if (!ix->plant_sorted) {
        size_t n = ix->plant_n;
        uint32_t* idx = (uint32_t*)malloc(n * sizeof(uint32_t));
        for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
        g_sort_keys = ix->plant_key;
        qsort(idx, n, sizeof(uint32_t), cmp_plant);
        uint64_t* k2 = (uint64_t*)malloc(n * sizeof(uint64_t));
        uint32_t* b2 = (uint32_t*)malloc(n * sizeof(uint32_t));
        for (size_t i = 0; i < n; ++i) { k2[i] = ix->plant_key[idx[i]]; b2[i] = ix->plant_bit[idx[i]]; }
        free(ix->plant_key); free(ix->plant_bit); free(idx);
        ix->plant_key = k2; ix->plant_bit = b2; ix->plant_cap = n;
        __atomic_store_n(&ix->plant_sorted, 1, __ATOMIC_RELEASE);
    }
    pthread_mutex_unlock(&g_plant_mu);
}

/* OR the planted bits of (page, row) that fall into row bytes [begin, begin + size) into dst */
static void plant_apply(const oracle_index* ix, uint32_t page, uint64_t row, uint64_t begin, uint64_t size, uint8_t* dst) {
    if (ix->plant_n == 0) return;
    if (!__atomic_load_n(&ix->plant_sorted, __ATOMIC_ACQUIRE)) plant_sort((oracle_index*)ix);
    uint64_t key = ((uint64_t)page << 44) | row;
    size_t lo = 0, hi = ix->plant_n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (ix->plant_key[mid] < key) lo = mid + 1; else hi = mid; }
    for (; lo < ix->plant_n && ix->plant_key[lo] == key; ++lo) {
        uint64_t byte = ix->plant_bit[lo] / 8;
        if (byte >= begin && byte < begin + size) dst[byte - begin] |= (uint8_t)(1u << (ix->plant_bit[lo] & 7u));
    }
}

/* ------------------------------------------------------------------------ */
/* read_from_disk -- classic_index/mmap_search_file.cpp:27-40 and
 * compact_index/mmap_search_file.cpp:34-67                                   */

static void gather_rows(const oracle_index* ix, const uint64_t* hashes, size_t nh,
                        uint8_t* rows, size_t begin, size_t size, size_t buffer_size) {
    if (ix->kind == 0) {
        uint64_t rsz = ix->page_size;            /* classic row_size */
        for (size_t i = 0; i < nh; ++i) {
            uint64_t row = hashes[i] % ix->sig[0];
            uint8_t* dst = rows + i * buffer_size;
            if (ix->synthetic) {
                oracle_synth_fill(0, ix->seed, 0, 1, ix->num_docs, 0, row, begin, size, dst);
                plant_apply(ix, 0, row, begin, size, dst);
            } else
                memcpy(dst, ix->page_data[0] + begin + row * rsz, size);
        }
        return;
    }
    uint64_t ps = ix->page_size;
    size_t begin_page = begin / ps;
    size_t end_page = (begin + size + ps - 1) / ps;
    for (size_t i = 0; i < nh; ++i) {
        size_t j = 0;
        for (size_t p = begin_page; p < end_page; ++p, ++j) {
            uint64_t row = hashes[i] % ix->sig[p];
            uint8_t* dst = rows + i * buffer_size + j * ps;
            if (ix->synthetic) {
                oracle_synth_fill(1, ix->seed, ps, ix->num_pages, ix->num_docs, (uint32_t)p, row, 0, ps, dst);
                plant_apply(ix, (uint32_t)p, row, 0, ps, dst);
            } else
                memcpy(dst, ix->page_data[p] + row * ps, ps);
        }
    }
}

/* aggregate_rows -- classic_search.cpp:279-307 */
static void and_rows(uint64_t H, size_t nh, uint8_t* rows, size_t size, size_t buffer_size) {
    for (size_t i = 0; i < nh; i += H) {
        uint8_t* r0 = rows + i * buffer_size;
        for (uint64_t j = 1; j < H; ++j) {
            const uint8_t* rj = r0 + j * buffer_size;
            size_t k = 0;
            for (; k + 8 <= size; k += 8) {
                uint64_t a, b; memcpy(&a, r0 + k, 8); memcpy(&b, rj + k, 8);
                a &= b; memcpy(r0 + k, &a, 8);
            }
            for (; k < size; ++k) r0[k] &= rj[k];
        }
    }
}

/* expansion tables -- classic_search.cpp:512-641 (u8), :683-940 (u16/SSE2),
 * :985-1002 (u32/SSE2).  Built at start-up instead of written out.           */
static uint64_t g_exp8[256];
static uint16_t g_exp16[256][8] __attribute__((aligned(16)));
static uint32_t g_exp32[16][4] __attribute__((aligned(16)));
static pthread_once_t g_tab_once = PTHREAD_ONCE_INIT;

static void build_tables(void) {
    for (int b = 0; b < 256; ++b) {
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) {
            if (b & (1 << i)) v |= 1ULL << (8 * i);
            g_exp16[b][i] = (uint16_t)((b >> i) & 1);
        }
        g_exp8[b] = v;
    }
    for (int n = 0; n < 16; ++n)
        for (int i = 0; i < 4; ++i) g_exp32[n][i] = (uint32_t)((n >> i) & 1);
}

/* compute_counts_u8_64 -- classic_search.cpp:643-655 */
static void add_rows_u8(uint64_t H, size_t nh, uint8_t* scores, const uint8_t* rows,
                        size_t size, size_t buffer_size) {
    for (size_t i = 0; i < nh; i += H) {
        const uint8_t* r = rows + i * buffer_size;
        for (size_t k = 0; k < size; ++k) {
            uint64_t c; memcpy(&c, scores + 8 * k, 8);
            c += g_exp8[r[k]];
            memcpy(scores + 8 * k, &c, 8);
        }
    }
}
/* compute_counts_u16_128 -- classic_search.cpp:942-957 (saturating add) */
static void add_rows_u16(uint64_t H, size_t nh, uint16_t* scores, const uint8_t* rows,
                         size_t size, size_t buffer_size) {
    __m128i* c = (__m128i*)scores;
    for (size_t i = 0; i < nh; i += H) {
        const uint8_t* r = rows + i * buffer_size;
        for (size_t k = 0; k < size; ++k) {
            __m128i v = _mm_loadu_si128(c + k);
            v = _mm_adds_epu16(v, _mm_load_si128((const __m128i*)g_exp16[r[k]]));
            _mm_storeu_si128(c + k, v);
        }
    }
}
/* compute_counts_u32_128 -- classic_search.cpp:1004-1022 */
static void add_rows_u32(uint64_t H, size_t nh, uint32_t* scores, const uint8_t* rows,
                         size_t size, size_t buffer_size) {
    __m128i* c = (__m128i*)scores;
    for (size_t i = 0; i < nh; i += H) {
        const uint8_t* r = rows + i * buffer_size;
        for (size_t k = 0; k < size; ++k) {
            __m128i lo = _mm_loadu_si128(c + 2 * k), hi = _mm_loadu_si128(c + 2 * k + 1);
            lo = _mm_add_epi32(lo, _mm_load_si128((const __m128i*)g_exp32[r[k] & 0xF]));
            hi = _mm_add_epi32(hi, _mm_load_si128((const __m128i*)g_exp32[r[k] >> 4]));
            _mm_storeu_si128(c + 2 * k, lo); _mm_storeu_si128(c + 2 * k + 1, hi);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* search_index_file<Score> -- classic_search.cpp:309-401                    */

typedef struct {
    const oracle_index* ix;
    const uint64_t* hashes; size_t nh;
    void* scores; int width;
    size_t batch_docs, total_docs, nbatches;
    size_t next; pthread_mutex_t mu;
    int err;
} batch_job;

static int run_batch(batch_job* jb, size_t b, double t[5]) {
    const oracle_index* ix = jb->ix;
    size_t doc_begin = b * jb->batch_docs;
    size_t doc_end = doc_begin + jb->batch_docs;
    if (doc_end > jb->total_docs) doc_end = jb->total_docs;
    if (doc_begin % 8 != 0) return ORACLE_ERR_GEOMETRY;
    size_t begin = doc_begin / 8;
    size_t size = (doc_end - doc_begin + 7) / 8;
    size_t buffer_size = (size + 7) & ~(size_t)7;
    if (ix->kind == 1 && begin % ix->page_size != 0) return ORACLE_ERR_GEOMETRY;
    if (begin + size > ix_row_size(ix)) return ORACLE_ERR_GEOMETRY;

    uint8_t* rows = NULL;
    size_t rows_bytes = buffer_size * jb->nh;
    if (posix_memalign((void**)&rows, 4096, rows_bytes > 0 ? rows_bytes : 4096) != 0)
        return ORACLE_ERR_ARG;
    memset(rows, 0, buffer_size * jb->nh);                 /* allocate_aligned zero-fills: misc.hpp:42-50 */

    double t0 = now_s();
    gather_rows(ix, jb->hashes, jb->nh, rows, begin, size, buffer_size);
    double t1 = now_s();
    if (ix->num_hashes != 1) and_rows(ix->num_hashes, jb->nh, rows, size, buffer_size);
    double t2 = now_s();
    if (jb->width == 1) add_rows_u8(ix->num_hashes, jb->nh, (uint8_t*)jb->scores + 8 * begin, rows, size, buffer_size);
    else if (jb->width == 2) add_rows_u16(ix->num_hashes, jb->nh, (uint16_t*)jb->scores + 8 * begin, rows, size, buffer_size);
    else add_rows_u32(ix->num_hashes, jb->nh, (uint32_t*)jb->scores + 8 * begin, rows, size, buffer_size);
    double t3 = now_s();
    free(rows);
    t[T_IO] += t1 - t0; t[T_AND] += t2 - t1; t[T_ADD] += t3 - t2;
    return ORACLE_OK;
}

static void* batch_worker(void* arg) {
    batch_job* jb = (batch_job*)arg;
    double t[5] = { 0, 0, 0, 0, 0 };
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        size_t b = jb->next++;
        pthread_mutex_unlock(&jb->mu);
        if (b >= jb->nbatches) break;
        int rc = run_batch(jb, b, t);
        if (rc != ORACLE_OK) jb->err = rc;
    }
    timer_add(t);
    return NULL;
}

/* scores must be zeroed, length counts_size of this index, element size `width` */
static int search_index_file(const oracle_index* ix, const char* q, size_t qlen, void* scores,
                             int width, int threads, size_t* total_hashes) {
    uint64_t limit = width == 1 ? 0xFFu : width == 2 ? 0xFFFFu : 0xFFFFFFFFu;
    if (!(qlen - ix->term_size < limit))
        return fail(ORACLE_ERR_QUERY_TOO_LONG, "query too long%s", "");
    size_t T = qlen - ix->term_size + 1;
    size_t nh = T * ix->num_hashes;
    double t[5] = { 0, 0, 0, 0, 0 };
    double t0 = now_s();
    uint64_t* hashes = (uint64_t*)malloc(8 * (nh ? nh : 1));
    int rc = create_hashes(ix, q, qlen, hashes);
    t[T_HASH] = now_s() - t0;
    timer_add(t);
    if (rc != ORACLE_OK) { free(hashes); return rc; }
    *total_hashes += nh;

    /* classic_search.cpp:338-341 */
    uint64_t page_size = oracle_page_size(ix);
    size_t total = oracle_counts_size(ix);
    size_t batch = 128;
    if (8 * page_size > batch) batch = 8 * page_size;
    if (batch > total) batch = total;
    batch_job jb;
    memset(&jb, 0, sizeof jb);
    jb.ix = ix; jb.hashes = hashes; jb.nh = nh; jb.scores = scores; jb.width = width;
    jb.batch_docs = batch; jb.total_docs = total;
    jb.nbatches = batch ? (total + batch - 1) / batch : 0;
    pthread_mutex_init(&jb.mu, NULL);
    if (threads <= 1) {
        batch_worker(&jb);
    } else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
        for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, batch_worker, &jb);
        for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&jb.mu);
    free(hashes);
    if (jb.err) return fail(jb.err, "unsupported batch geometry (page_size vs 16-byte batches)%s", "");
    return ORACLE_OK;
}

/* Score type selection -- classic_search.cpp:453-504 */
static int pick_width(size_t qlen, uint32_t max_term) {
    size_t d = qlen - max_term;
    if (d < 0xFFu) return 1;
    if (d < 0xFFFFu) return 2;
    if (d < 0xFFFFFFFFu) return 4;
    return 0;
}

static uint32_t score_at(const void* s, int width, size_t i) {
    if (width == 1) return ((const uint8_t*)s)[i];
    if (width == 2) return ((const uint16_t*)s)[i];
    return ((const uint32_t*)s)[i];
}

int oracle_counts(oracle_index* ix, const char* query, size_t len, int threads,
                  uint32_t* counts, int* width_out) {
    pthread_once(&g_tab_once, build_tables);
    if (len < ix->term_size) return fail(ORACLE_ERR_QUERY_TOO_SHORT, "query too short%s", "");
    int width = pick_width(len, ix->term_size);
    if (!width) return fail(ORACLE_ERR_QUERY_TOO_LONG, "query too long%s", "");
    size_t n = oracle_counts_size(ix);
    void* scores = NULL;
    if (posix_memalign(&scores, 16, (n ? n : 16) * (size_t)width) != 0) return ORACLE_ERR_ARG;
    memset(scores, 0, n * (size_t)width);
    size_t th = 0;
    int rc = search_index_file(ix, query, len, scores, width, threads, &th);
    if (rc == ORACLE_OK) {
        for (size_t i = 0; i < n; ++i) counts[i] = score_at(scores, width, i);
        if (width_out) *width_out = width;
    }
    free(scores);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* counts_to_result -- classic_search.cpp:109-202 ; search -- :403-505       */

typedef struct { uint32_t score, index, doc; } hit;

/* order: score descending, then (index, doc) ascending */
static int hit_before(const hit* a, const hit* b) {
    if (a->score != b->score) return a->score > b->score;
    if (a->index != b->index) return a->index < b->index;
    return a->doc < b->doc;
}

static void sift_down(hit* h, size_t n, size_t i) {
    /* max-heap w.r.t. "comes later": root is the worst of the kept prefix */
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && hit_before(&h[m], &h[l])) m = l;
        if (r < n && hit_before(&h[m], &h[r])) m = r;
        if (m == i) return;
        hit t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

/* std::partial_sort semantics: first k of n in order (rest unspecified) */
static void partial_sort_hits(hit* h, size_t k, size_t n) {
    if (k == 0) return;
    for (size_t i = k / 2; i-- > 0;) sift_down(h, k, i);
    for (size_t i = k; i < n; ++i) {
        if (hit_before(&h[i], &h[0])) {
            hit t = h[0]; h[0] = h[i]; h[i] = t;
            sift_down(h, k, 0);
        }
    }
    for (size_t e = k; e > 1; --e) {
        hit t = h[0]; h[0] = h[e - 1]; h[e - 1] = t;
        sift_down(h, e - 1, 0);
    }
}

int oracle_search(oracle_index* const* ixs, size_t n, const char* query, size_t len,
                  double threshold, size_t num_results, int threads,
                  uint32_t* out_index, uint32_t* out_doc, uint32_t* out_score,
                  size_t cap, size_t* n_out) {
    pthread_once(&g_tab_once, build_tables);
    *n_out = 0;
    if (n == 0) return ORACLE_OK;
    size_t* sum = (size_t*)calloc(n + 1, sizeof(size_t));
    uint32_t max_term = 0;
    for (size_t i = 0; i < n; ++i) {
        sum[i + 1] = sum[i] + oracle_counts_size(ixs[i]);
        if (ixs[i]->term_size > max_term) max_term = ixs[i]->term_size;
    }
    if (len < max_term) { free(sum); return fail(ORACLE_ERR_QUERY_TOO_SHORT, "query too short%s", ""); }
    size_t total = sum[n];
    size_t* thr = (size_t*)malloc(sizeof(size_t) * n);
    for (size_t i = 0; i < n; ++i)
        thr[i] = (size_t)ceil(threshold * (double)(len - ixs[i]->term_size + 1));
    num_results = num_results == 0 ? total : (num_results < total ? num_results : total);

    int width = pick_width(len, max_term);
    if (!width) { free(sum); free(thr); return fail(ORACLE_ERR_QUERY_TOO_LONG, "query too long%s", ""); }
    void* scores = NULL;
    if (posix_memalign(&scores, 16, (total ? total : 16) * (size_t)width) != 0) { free(sum); free(thr); return ORACLE_ERR_ARG; }
    memset(scores, 0, total * (size_t)width);
    size_t total_hashes = 0;
    int rc = ORACLE_OK;
    for (size_t i = 0; i < n && rc == ORACLE_OK; ++i)
        rc = search_index_file(ixs[i], query, len, (uint8_t*)scores + sum[i] * (size_t)width,
                               width, threads, &total_hashes);
    if (rc != ORACLE_OK) { free(scores); free(sum); free(thr); return rc; }

    double t0 = now_s();
    hit* hits = (hit*)malloc(sizeof(hit) * (total ? total : 1));
    size_t nhit = 0;
    for (size_t k = 0; k < n; ++k)
        for (size_t d = 0; d < ixs[k]->num_docs; ++d) {
            uint32_t s = score_at(scores, width, sum[k] + d);
            if (s >= thr[k]) { hits[nhit].score = s; hits[nhit].index = (uint32_t)k; hits[nhit].doc = (uint32_t)d; nhit++; }
        }
    if (num_results > nhit) num_results = nhit;
    if (total_hashes > 1) partial_sort_hits(hits, num_results, nhit);   /* max_counts > 1 */
    if (num_results > cap) num_results = cap;
    for (size_t i = 0; i < num_results; ++i) {
        out_index[i] = hits[i].index; out_doc[i] = hits[i].doc; out_score[i] = hits[i].score;
    }
    *n_out = num_results;
    double t[5] = { 0, 0, 0, 0, now_s() - t0 };
    timer_add(t);
    free(hits); free(scores); free(sum); free(thr);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* timing loop for the CPU baseline: runs oracle_search over queries
 * 0, 1, ... until `seconds` have elapsed (at least one query), entirely in C
 * so that no marshalling cost is included.  Returns the number of queries run. */
size_t oracle_search_many(oracle_index* const* ixs, size_t n, const char* text,
                          const uint64_t* offsets, size_t nq, double threshold,
                          size_t num_results, int threads, double seconds,
                          double* elapsed_out, uint64_t* checksum_out) {
    size_t cap = 0;
    for (size_t i = 0; i < n; ++i) cap += oracle_counts_size(ixs[i]);
    uint32_t* oi = (uint32_t*)malloc(4 * (cap ? cap : 1));
    uint32_t* od = (uint32_t*)malloc(4 * (cap ? cap : 1));
    uint32_t* os = (uint32_t*)malloc(4 * (cap ? cap : 1));
    uint64_t sum = 0;
    size_t done = 0;
    double t0 = now_s();
    while (done < nq) {
        size_t nout = 0;
        int rc;
        if (threshold < 0) {
            /* counts only: the score_list of classic_search.cpp:456-467, no counts_to_result */
            size_t off = 0;
            rc = ORACLE_OK;
            for (size_t i = 0; i < n && rc == ORACLE_OK; ++i) {
                rc = oracle_counts(ixs[i], text + offsets[done], (size_t)(offsets[done + 1] - offsets[done]),
                                   threads, os + off, NULL);
                off += oracle_counts_size(ixs[i]);
            }
            nout = cap < 4 ? cap : 4;
        } else {
            rc = oracle_search(ixs, n, text + offsets[done], (size_t)(offsets[done + 1] - offsets[done]),
                               threshold, num_results, threads, oi, od, os, cap, &nout);
        }
        if (rc != ORACLE_OK) break;
        for (size_t k = 0; k < nout && k < 4; ++k) sum += (uint64_t)os[k] * 31;
        done++;
        if (now_s() - t0 >= seconds) break;
    }
    if (elapsed_out) *elapsed_out = now_s() - t0;
    if (checksum_out) *checksum_out = sum;
    free(oi); free(od); free(os);
    return done;
}
