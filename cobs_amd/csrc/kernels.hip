// cobs_amd/csrc/kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the
// COBS query path.  Nothing here is translated from the reference: its CPU code
// gathers rows into a scratch buffer and expands every row BYTE into eight
// counter lanes through a lookup table (reference cobs/query/classic_search.cpp:
// 643-1022); at HBM speed that is VALU-bound.  Here each lane owns a 16-byte
// column chunk (128 documents) of the bit-sliced matrix and keeps the
// per-document counters bit-sliced as well ("vertical counters"): NP bit planes
// per 32-bit column word, updated with a Harley-Seal carry-save adder tree, so a
// gathered row costs ~4.4 VALU ops per 32 documents and the kernel stays on the
// HBM roofline.
//
//   K1 hash_kernel    canonicalise + XXH64 + (hash % S_p) per sub-index
//                     (create_hashes, classic_search.cpp:66-107; canonicalize_kmer,
//                      util/query.cpp:143-199; modulo at
//                      classic_index/mmap_search_file.cpp:35 and
//                      compact_index/mmap_search_file.cpp:58)
//   K2 scan_kernel    row gather + AND over the H hash rows + per-document count
//                     (+ optional threshold selection)
//                     (read_from_disk, aggregate_rows :279-307, compute_counts
//                      :643-1022, threshold filter of counts_to_result :127-132)
//   K3 topk_kernel    exact top-k per query (partial_sort of counts_to_result, :134-145)
//   build_kernel      index construction (classic_index.cpp:40-73): hash terms, set document bits
//   synth_kernel / repitch_kernel   index staging helpers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "device_types.hpp"
#include "kernels.hpp"

namespace cobs_amd {

// ---------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// exact n % d with the precomputed m = floor((2^64-1)/d); the estimate
// q = hi64(n*m) is at most 2 below the true quotient.
__device__ __forceinline__ uint64_t fast_mod(uint64_t n, uint64_t d, uint64_t m) {
    uint64_t q = __umul64hi(n, m);
    uint64_t r = n - q * d;
    if (r >= d) r -= d;
    if (r >= d) r -= d;
    return r;
}

__device__ __forceinline__ uint32_t fwd_base(uint32_t c) {
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 0u;
}
__device__ __forceinline__ uint32_t rev_base(uint32_t c) {
    return c == 'A' ? (uint32_t)'T' : c == 'C' ? (uint32_t)'G' : c == 'G' ? (uint32_t)'C'
         : c == 'T' ? (uint32_t)'A' : 0u;
}

// ---------------------------------------------------------------------------
// K1: one thread per query position.

constexpr uint64_t XP1 = 0x9E3779B185EBCA87ULL;
constexpr uint64_t XP2 = 0xC2B2AE3D27D4EB4FULL;
constexpr uint64_t XP3 = 0x165667B19E3779F9ULL;
constexpr uint64_t XP4 = 0x85EBCA77C2B2AE63ULL;
constexpr uint64_t XP5 = 0x27D4EB2F165667C5ULL;

__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) {
    return rotl64(acc + in * XP2, 31) * XP1;
}
__device__ __forceinline__ uint64_t xmerge(uint64_t h, uint64_t v) {
    return (h ^ xround(0, v)) * XP1 + XP4;
}

// canonical k-mer as a byte accessor: mode 0 raw, 1 forward-mapped, 2 reverse complement
struct KmerView {
    const uint8_t* p;
    uint32_t k;
    uint32_t mode;
    __device__ __forceinline__ uint32_t at(uint32_t i) const {
        if (mode == 0) return p[i];
        if (mode == 1) return fwd_base(p[i]);
        return rev_base(p[k - 1 - i]);
    }
    __device__ __forceinline__ uint64_t le64(uint32_t i) const {
        uint64_t v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) v |= (uint64_t)at(i + b) << (8 * b);
        return v;
    }
    __device__ __forceinline__ uint64_t le32(uint32_t i) const {
        uint64_t v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) v |= (uint64_t)at(i + b) << (8 * b);
        return v;
    }
};

// XXH64 of the viewed k bytes (public xxHash specification; any k)
__device__ uint64_t xxh64_view(const KmerView& kv, uint64_t seed) {
    const uint32_t len = kv.k;
    uint32_t pos = 0;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        do {
            v1 = xround(v1, kv.le64(pos));
            v2 = xround(v2, kv.le64(pos + 8));
            v3 = xround(v3, kv.le64(pos + 16));
            v4 = xround(v4, kv.le64(pos + 24));
            pos += 32;
        } while (pos + 32 <= len);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)len;
    while (pos + 8 <= len) {
        h ^= xround(0, kv.le64(pos));
        h = rotl64(h, 27) * XP1 + XP4;
        pos += 8;
    }
    if (pos + 4 <= len) {
        h ^= kv.le32(pos) * XP1;
        h = rotl64(h, 23) * XP2 + XP3;
        pos += 4;
    }
    while (pos < len) {
        h ^= (uint64_t)kv.at(pos) * XP5;
        h = rotl64(h, 11) * XP1;
        pos++;
    }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

template <typename IdxT>
__global__ __launch_bounds__(256) void hash_kernel(HashArgs a, uint64_t total_threads) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the grid may be larger than the batch needs (a captured launch is replayed for other query lengths)
    if (gid >= total_threads || gid >= a.span_off[a.nq]) return;
    // query of this thread: last q with span_off[q] <= gid
    uint32_t lo = 0, hi = a.nq;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (a.span_off[mid] <= gid) lo = mid; else hi = mid;
    }
    const uint32_t q = lo;
    const uint64_t qbase = a.span_off[q];
    const uint32_t i = (uint32_t)(gid - qbase);
    const uint32_t len = a.q_len[q];
    const uint8_t* text = a.text + qbase;
    const uint32_t k = a.term_size;

    // canonicalize == 1: any character outside ACGT makes the query invalid
    // (the reference dies, classic_search.cpp:93-96).  Every character of a
    // query of length >= k lies in some k-mer.
    if (a.canonicalize != 0 && i < len) {
        if (fwd_base(text[i]) == 0) atomicMax(a.err_query, 0xFFFFFFFFu - q);   // first bad query wins
    }

    const uint64_t b0 = a.blk_off[q];
    const uint32_t nblk = (uint32_t)(a.blk_off[q + 1] - b0);
    const uint32_t T = len - k + 1;
    // every (query, sub-index) table has nblk blocks of 8 terms plus one all-padding
    // block that lanes without work in a trip of K2 point at
    const uint32_t tblk = nblk + 1u;
    if (i >= tblk * 8u) return;
    const uint32_t H = a.num_hashes;
    const uint32_t blk = i >> 3, sub = i & 7u;
    IdxT* out = reinterpret_cast<IdxT*>(a.table) + ((b0 + q) * a.npages) * (8ull * H);

    if (i >= T) {       // padding term: the all-zero row of every sub-index
        for (uint32_t p = 0; p < a.npages; ++p) {
            const IdxT zr = (IdxT)a.pages[p].sig;
            IdxT* o = out + ((uint64_t)p * tblk + blk) * (8ull * H) + sub;
            for (uint32_t j = 0; j < H; ++j) o[j * 8] = zr;
        }
        return;
    }

    KmerView kv{text + i, k, 0u};
    if (a.canonicalize != 0) {
        // util/query.cpp:143-199: first strict difference between the forward
        // base and the complement of the mirrored base decides; the middle base
        // of an odd k is not compared; ties keep the forward k-mer.
        uint32_t mode = 1;
        for (uint32_t s = 0; s < k / 2; ++s) {
            const int f = (int)fwd_base(text[i + s]);
            const int r = (int)rev_base(text[i + k - 1 - s]);
            if (f < r) break;
            if (f > r) { mode = 2; break; }
        }
        kv.mode = mode;
    }
    for (uint32_t j = 0; j < H; ++j) {
        const uint64_t h = xxh64_view(kv, (uint64_t)j);
        for (uint32_t p = 0; p < a.npages; ++p) {
            const PageDev pg = a.pages[p];
            out[((uint64_t)p * tblk + blk) * (8ull * H) + j * 8 + sub] = (IdxT)fast_mod(h, pg.sig, pg.magic);
        }
    }
}

// K1 specialised for k = 31 (the COBS default): the 31-mer lives in eight 32-bit
// registers, complement and reversal are done four bases at a time, and the
// reference's comparison of the first 15 positions (util/query.cpp:155-190) becomes
// a big-endian integer comparison of forward vs reverse complement.
__device__ __forceinline__ uint32_t comp4(uint32_t w) {
    // A(0x41)<->T(0x54): xor 0x15, C(0x43)<->G(0x47): xor 0x04; C and G have bit 1 set
    const uint32_t m = (w >> 1) & 0x01010101u;
    return w ^ 0x15151515u ^ (m | (m << 4));
}

__device__ __forceinline__ uint64_t xxh64_31(const uint32_t (&c)[8], uint64_t seed) {
    // public XXH64 spec for len = 31 < 32: 3 x 8 bytes, 1 x 4 bytes, 3 x 1 byte
    uint64_t h = seed + XP5 + 31ull;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint64_t v = (uint64_t)c[2 * i] | ((uint64_t)c[2 * i + 1] << 32);
        h ^= xround(0, v);
        h = rotl64(h, 27) * XP1 + XP4;
    }
    h ^= (uint64_t)c[6] * XP1;
    h = rotl64(h, 23) * XP2 + XP3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        h ^= (uint64_t)((c[7] >> (8 * i)) & 0xFFu) * XP5;
        h = rotl64(h, 11) * XP1;
    }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

// canonical form of a 31-mer of valid bases held in eight dwords (byte 31 zero): c = f or its
// reverse complement (util/query.cpp:143-199)
__device__ __forceinline__ void canon31(const uint32_t (&f)[8], uint32_t (&c)[8]) {
    // reverse complement: B[j] = comp(raw[31 - j]) for the 32-byte block, then drop B[0]
    uint32_t rv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rv[j] = __builtin_bswap32(comp4(f[7 - j]));
    uint32_t rc[8];
#pragma unroll
    for (int j = 0; j < 7; ++j) rc[j] = (rv[j] >> 8) | (rv[j + 1] << 24);
    rc[7] = rv[7] >> 8;
    // first strict difference among positions 0..14 decides (big-endian compare);
    // the middle base (position 15) is never compared; ties keep the forward k-mer
    bool use_rc = false, decided = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t x = __builtin_bswap32(f[j]), y = __builtin_bswap32(rc[j]);
        if (j == 3) { x >>= 8; y >>= 8; }
        if (!decided && x != y) { use_rc = x > y; decided = true; }
    }
    if (use_rc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = rc[j];
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void hash_kernel_k31(HashArgs a, uint64_t total_threads) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_threads || gid >= a.span_off[a.nq]) return;
    uint32_t lo = 0, hi = a.nq;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.span_off[mid] <= gid) lo = mid; else hi = mid;
    }
    const uint32_t q = lo;
    const uint64_t qbase = a.span_off[q];
    const uint32_t i = (uint32_t)(gid - qbase);
    const uint32_t len = a.q_len[q];
    const uint8_t* text = a.text + qbase;
    if (a.canonicalize != 0 && i < len) {
        if (fwd_base(text[i]) == 0) atomicMax(a.err_query, 0xFFFFFFFFu - q);   // first bad query wins
    }
    const uint64_t b0 = a.blk_off[q];
    const uint32_t nblk = (uint32_t)(a.blk_off[q + 1] - b0);
    const uint32_t T = len - 31u + 1u;
    const uint32_t tblk = nblk + 1u;
    if (i >= tblk * 8u) return;
    const uint32_t H = a.num_hashes;
    const uint32_t blk = i >> 3, sub = i & 7u;
    IdxT* out = reinterpret_cast<IdxT*>(a.table) + ((b0 + q) * a.npages) * (8ull * H);
    if (i >= T) {
        for (uint32_t p = 0; p < a.npages; ++p) {
            const IdxT zr = (IdxT)a.pages[p].sig;
            IdxT* o = out + ((uint64_t)p * tblk + blk) * (8ull * H) + sub;
            for (uint32_t j = 0; j < H; ++j) o[j * 8] = zr;
        }
        return;
    }
    // the k-mer and one following byte as 8 (unaligned) dwords; the text buffer is padded
    uint32_t f[8];
    {
        const uint8_t* p = text + i;
        const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(p - mis);
        uint32_t r[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) r[j] = w[j];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            f[j] = mis == 0 ? r[j] : (uint32_t)(((uint64_t)r[j] | ((uint64_t)r[j + 1] << 32)) >> (8 * mis));
    }
    f[7] &= 0x00FFFFFFu;                          // byte 31 is not part of the 31-mer
    uint32_t c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = f[j];
    if (a.canonicalize != 0) canon31(f, c);
    for (uint32_t j = 0; j < H; ++j) {
        const uint64_t h = xxh64_31(c, (uint64_t)j);
        for (uint32_t p = 0; p < a.npages; ++p) {
            const PageDev pg = a.pages[p];
            out[((uint64_t)p * tblk + blk) * (8ull * H) + j * 8 + sub] = (IdxT)fast_mod(h, pg.sig, pg.magic);
        }
    }
}

// ---------------------------------------------------------------------------
// K2: gather + AND + bit-sliced count.
//
// Work-group = (query q, tile of W sixteen-byte column chunks, W = 4..64); its NW
// waves x 64/W lane groups split the query's 8-term blocks round-robin and merge
// their partial plane counters (shuffles inside a wave, LDS across waves) at the
// end.  A lane owns one chunk: 128 documents, held as 4 column words x NP bit planes.

// carry-save adder: (h, l) = a + b + c per bit position.  gfx950 has a
// three-input boolean op (v_bitop3_b32, 8-bit truth table), so majority (0xE8)
// and parity (0x96) are one instruction each: a CSA is 2 VALU ops.
__device__ __forceinline__ void csa(uint32_t& h, uint32_t& l, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t hh = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
    const uint32_t ll = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    h = hh;
    l = ll;
}

// fold eight gathered row words into planes 0..2 of one column word; returns the
// carry into plane 3 ("eights")
template <int NP>
__device__ __forceinline__ uint32_t absorb8(uint32_t (&pl)[NP], uint32_t x0, uint32_t x1, uint32_t x2,
                                            uint32_t x3, uint32_t x4, uint32_t x5, uint32_t x6,
                                            uint32_t x7) {
    uint32_t t2a, t2b, f4a, f4b, e8;
    csa(t2a, pl[0], pl[0], x0, x1);
    csa(t2b, pl[0], pl[0], x2, x3);
    csa(f4a, pl[1], pl[1], t2a, t2b);
    csa(t2a, pl[0], pl[0], x4, x5);
    csa(t2b, pl[0], pl[0], x6, x7);
    csa(f4b, pl[1], pl[1], t2a, t2b);
    csa(e8, pl[2], pl[2], f4a, f4b);
    return e8;
}

// ripple a carry into planes FROM..NP-1
template <int NP, int FROM>
__device__ __forceinline__ void ripple(uint32_t (&pl)[NP], uint32_t carry) {
#pragma unroll
    for (int k = FROM; k < NP; ++k) {
        const uint32_t t = pl[k] & carry;
        pl[k] ^= carry;
        carry = t;
    }
}

// eights[w] = carry-outs of one 8-row block for the lane's 4 column words
template <int NP>
__device__ __forceinline__ void absorb_block(uint32_t (&pl)[4][NP], const uint4 (&X)[8],
                                             uint32_t (&e8)[4]) {
    e8[0] = absorb8<NP>(pl[0], X[0].x, X[1].x, X[2].x, X[3].x, X[4].x, X[5].x, X[6].x, X[7].x);
    e8[1] = absorb8<NP>(pl[1], X[0].y, X[1].y, X[2].y, X[3].y, X[4].y, X[5].y, X[6].y, X[7].y);
    e8[2] = absorb8<NP>(pl[2], X[0].z, X[1].z, X[2].z, X[3].z, X[4].z, X[5].z, X[6].z, X[7].z);
    e8[3] = absorb8<NP>(pl[3], X[0].w, X[1].w, X[2].w, X[3].w, X[4].w, X[5].w, X[6].w, X[7].w);
}

// one block on its own: its eights go straight into plane 3 and up
template <int NP>
__device__ __forceinline__ void retire_single(uint32_t (&pl)[4][NP], const uint32_t (&e8)[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) ripple<NP, 3>(pl[w], e8[w]);
}

// two blocks: add both eights into plane 3 with one more CSA, ripple the sixteens
template <int NP>
__device__ __forceinline__ void retire_pair(uint32_t (&pl)[4][NP], const uint32_t (&ea)[4],
                                            const uint32_t (&eb)[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t s16;
        csa(s16, pl[w][3], pl[w][3], ea[w], eb[w]);
        ripple<NP, 4>(pl[w], s16);
    }
}

// Half a block (generic-H row loop, round 6): four row words into planes 0..1 of one column word; the first half
// leaves its carry into plane 2 ("fours") pending, the second half adds both fours into plane 2 and returns the eights.
template <int NP>
__device__ __forceinline__ uint32_t absorb4(uint32_t (&pl)[NP], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    uint32_t t2a, t2b, f4;
    csa(t2a, pl[0], pl[0], x0, x1);
    csa(t2b, pl[0], pl[0], x2, x3);
    csa(f4, pl[1], pl[1], t2a, t2b);
    return f4;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ uint4 load_row(const uint8_t* lane_base, uint64_t row, uint32_t pitch) {
    const u32x4* p = reinterpret_cast<const u32x4*>(lane_base + row * pitch);
    const u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Eight row indices of one (block, hash): 32-bit (two 16-byte loads) or, for sub-indexes with
// 2^32 or more rows (signature sizes of human-sized documents), 64-bit (four loads).
template <typename IdxT> struct Idx8;
template <> struct Idx8<uint32_t> { uint4 a, b; };
template <> struct Idx8<uint64_t> { uint4 a, b, c, d; };

__device__ __forceinline__ Idx8<uint32_t> load_idx8(const uint32_t* p) {
    const uint4* t = reinterpret_cast<const uint4*>(p);
    return Idx8<uint32_t>{t[0], t[1]};
}
__device__ __forceinline__ Idx8<uint64_t> load_idx8(const uint64_t* p) {
    const uint4* t = reinterpret_cast<const uint4*>(p);
    return Idx8<uint64_t>{t[0], t[1], t[2], t[3]};
}

template <bool NT>
__device__ __forceinline__ void issue_rows(uint4 (&X)[8], const uint8_t* lane_base, uint32_t pitch,
                                           const Idx8<uint32_t>& i) {
    X[0] = load_row<NT>(lane_base, i.a.x, pitch);
    X[1] = load_row<NT>(lane_base, i.a.y, pitch);
    X[2] = load_row<NT>(lane_base, i.a.z, pitch);
    X[3] = load_row<NT>(lane_base, i.a.w, pitch);
    X[4] = load_row<NT>(lane_base, i.b.x, pitch);
    X[5] = load_row<NT>(lane_base, i.b.y, pitch);
    X[6] = load_row<NT>(lane_base, i.b.z, pitch);
    X[7] = load_row<NT>(lane_base, i.b.w, pitch);
}

template <bool NT>
__device__ __forceinline__ void issue_rows4(uint4 (&X)[4], const uint8_t* lane_base, uint32_t pitch, const uint4& i) {
    X[0] = load_row<NT>(lane_base, i.x, pitch);
    X[1] = load_row<NT>(lane_base, i.y, pitch);
    X[2] = load_row<NT>(lane_base, i.z, pitch);
    X[3] = load_row<NT>(lane_base, i.w, pitch);
}

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return (uint64_t)hi << 32 | lo; }

template <bool NT>
__device__ __forceinline__ void issue_rows(uint4 (&X)[8], const uint8_t* lane_base, uint32_t pitch,
                                           const Idx8<uint64_t>& i) {
    X[0] = load_row<NT>(lane_base, u64_of(i.a.x, i.a.y), pitch);
    X[1] = load_row<NT>(lane_base, u64_of(i.a.z, i.a.w), pitch);
    X[2] = load_row<NT>(lane_base, u64_of(i.b.x, i.b.y), pitch);
    X[3] = load_row<NT>(lane_base, u64_of(i.b.z, i.b.w), pitch);
    X[4] = load_row<NT>(lane_base, u64_of(i.c.x, i.c.y), pitch);
    X[5] = load_row<NT>(lane_base, u64_of(i.c.z, i.c.w), pitch);
    X[6] = load_row<NT>(lane_base, u64_of(i.d.x, i.d.y), pitch);
    X[7] = load_row<NT>(lane_base, u64_of(i.d.z, i.d.w), pitch);
}

__device__ __forceinline__ void and_rows(uint4 (&X)[8], const uint4 (&Y)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        X[t].x &= Y[t].x; X[t].y &= Y[t].y; X[t].z &= Y[t].z; X[t].w &= Y[t].w;
    }
}

// ---- LDS-staged variant of the row pipeline (LDSS): rows travel HBM -> LDS (direct-to-LDS DMA,
// global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per instruction) -> VGPR (ds_read_b128)
// instead of HBM -> VGPR.  hipcc neither counts asm memory operations nor pipelines LDS-DMA
// (it would drain it with vmcnt(0)), so the loop's VMEM operations are inline asm with counted
// s_waitcnt vmcnt(N); the waits name the registers they make valid so that the compiler
// cannot touch them earlier.  Kept as a measured A/B against the VGPR-direct pipeline
// (scripts/ab.py ... lds=1; result in profiles/).
__device__ __forceinline__ void asm_load_idx(u32x4& a, u32x4& b, const uint32_t* p) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
// one row: the wave's 64 lanes x 16 bytes land at LDS address lds_dst + lane * 16
__device__ __forceinline__ void asm_glds16(const uint8_t* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_rows(const uint8_t* lane_base, uint32_t pitch, const u32x4& a, const u32x4& b,
                                          uint32_t lds_dst) {
    asm_glds16(lane_base + (uint64_t)a.x * pitch, lds_dst);
    asm_glds16(lane_base + (uint64_t)a.y * pitch, lds_dst + 1024u);
    asm_glds16(lane_base + (uint64_t)a.z * pitch, lds_dst + 2048u);
    asm_glds16(lane_base + (uint64_t)a.w * pitch, lds_dst + 3072u);
    asm_glds16(lane_base + (uint64_t)b.x * pitch, lds_dst + 4096u);
    asm_glds16(lane_base + (uint64_t)b.y * pitch, lds_dst + 5120u);
    asm_glds16(lane_base + (uint64_t)b.z * pitch, lds_dst + 6144u);
    asm_glds16(lane_base + (uint64_t)b.w * pitch, lds_dst + 7168u);
}
#define COBS_WAIT_VM_IDX(N, A, B) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(A), "+v"(B) : : "memory")
#define COBS_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" : : : "memory")

// Phase timestamps of sampled work-groups (tuning builds only: make -C cobs_amd/csrc timing;
// scripts/phase_times.py).  COBS_STAMP(k) drains the memory queues first so that the time of a
// phase includes the loads it issued.
#ifdef COBS_SCAN_TIMING
#define COBS_STAMP(K)                                                                           \
    do {                                                                                        \
        if (a.dbg && (blockIdx.x % a.dbg_every) == 0u && blockIdx.x / a.dbg_every < a.dbg_slots) { \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                         \
            const uint64_t t_ = __builtin_amdgcn_s_memtime();                                   \
            if (lane == 0u) a.dbg[((uint64_t)(blockIdx.x / a.dbg_every) * 4u + wave) * 8u + (K)] = t_; \
        }                                                                                       \
    } while (0)
#else
#define COBS_STAMP(K) do { } while (0)
#endif

// LDS bytes in front of the expansion table / tile metadata: the merge buffers of the waves
// ([NW/2 (min 1)][NP][64 lanes] of uint4), the row staging ring of the LDS-staged variant, or --
// 8-bit scores -- the score staging buffer of the epilogue, whichever is largest
template <int NP, int NW, size_t OutBytes, bool LDSS>
__host__ __device__ constexpr size_t scan_lds_front() {
    size_t f = LDSS ? (size_t)NW * 16384 : (size_t)(NW >= 2 ? NW / 2 : 1) * NP * 64 * sizeof(uint4);
    if (OutBytes == 1 && f < 64 * 144) f = 64 * 144;
    return f;
}

// 32 documents x NP (<= 8) bit planes -> the documents' 8-bit counts, in registers: byte b of every
// plane is gathered with v_perm_b32 (two 4x4 byte transposes), then an 8x8 bit transpose (three
// delta swaps) turns "bit i of byte k = bit k of document 8b+i" into one byte per document.
// out[j] = counts of documents 4j .. 4j+3.  24 VALU per 8 documents, no LDS table.
template <int NP>
__device__ __forceinline__ void planes_to_bytes32(const uint32_t (&P)[NP], uint32_t (&out)[8]) {
    static_assert(NP <= 8, "8-bit scores hold at most 8 planes");
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = k < NP ? P[k] : 0u;
    uint32_t x[4], y[4];
    {
        const uint32_t t0 = __builtin_amdgcn_perm(q[1], q[0], 0x05010400u), t1 = __builtin_amdgcn_perm(q[1], q[0], 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(q[3], q[2], 0x05010400u), t3 = __builtin_amdgcn_perm(q[3], q[2], 0x07030602u);
        x[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); x[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
        x[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); x[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
    }
    {
        const uint32_t t0 = __builtin_amdgcn_perm(q[5], q[4], 0x05010400u), t1 = __builtin_amdgcn_perm(q[5], q[4], 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(q[7], q[6], 0x05010400u), t3 = __builtin_amdgcn_perm(q[7], q[6], 0x07030602u);
        y[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); y[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
        y[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); y[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        uint32_t lo = x[b], hi = y[b], t;
        t = (lo ^ (lo >> 7)) & 0x00AA00AAu; lo = lo ^ t ^ (t << 7);
        t = (hi ^ (hi >> 7)) & 0x00AA00AAu; hi = hi ^ t ^ (t << 7);
        t = (lo ^ (lo >> 14)) & 0x0000CCCCu; lo = lo ^ t ^ (t << 14);
        t = (hi ^ (hi >> 14)) & 0x0000CCCCu; hi = hi ^ t ^ (t << 14);
        out[2 * b] = (lo & 0x0F0F0F0Fu) | ((hi << 4) & 0xF0F0F0F0u);
        out[2 * b + 1] = ((lo >> 4) & 0x0F0F0F0Fu) | (hi & 0xF0F0F0F0u);
    }
}
// LDS bytes between the score blocks of two lanes in the staging buffer of the 8-bit epilogue:
// 128 bytes of scores + 16 of padding, so that the 16-byte writes of 16 lanes hit 16 different banks
constexpr uint32_t kStageStride = 144u;

// Cross-lane steps inside a row of 16 lanes as DPP modifiers of the VALU (v_add_u32 ..._dpp: no trip through the
// LDS crossbar, which a ds_bpermute-based __shfl costs -- ~100 cycles of latency per dependent step).
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND_ZERO = true>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND_ZERO);
}
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowMirror = 0x140;     // lane i <- lane 15 - i of its row
constexpr int kDppHalfMirror = 0x141;    // lane i <- lane 7 - i of its half row
constexpr int kDppRowShr = 0x110;        // + n: lane i <- lane i - n of its row (0 shifted in)
constexpr int kDppBcast15 = 0x142;       // lane 15 of a row -> every lane of the next row
constexpr int kDppBcast31 = 0x143;       // lane 31 -> every lane of rows 2 and 3

// sum of `c` over the W (power of two, 4..64) consecutive lanes of a lane group, in every lane of the group
__device__ __forceinline__ uint32_t group_allsum(uint32_t c, uint32_t W) {
    c += dpp_mov<kDppQuadXor1>(c);
    c += dpp_mov<kDppQuadXor2>(c);                      // every lane: the sum of its quad
    if (W >= 8u) c += dpp_mov<kDppHalfMirror>(c);       // + the other quad of the half row
    if (W >= 16u) c += dpp_mov<kDppRowMirror>(c);       // + the other half of the row
    if (W >= 32u) c += (uint32_t)__shfl_xor(c, 16);
    if (W >= 64u) c += (uint32_t)__shfl_xor(c, 32);
    return c;
}

// inclusive prefix sum of `v` over the lanes of a lane group (col = lane inside the group)
__device__ __forceinline__ uint32_t group_incl_scan(uint32_t v, uint32_t W, uint32_t col) {
    uint32_t t;
    t = dpp_mov<kDppRowShr + 1>(v); v += col >= 1u ? t : 0u;
    t = dpp_mov<kDppRowShr + 2>(v); v += col >= 2u ? t : 0u;
    if (W >= 8u) { t = dpp_mov<kDppRowShr + 4>(v); v += col >= 4u ? t : 0u; }
    if (W >= 16u) { t = dpp_mov<kDppRowShr + 8>(v); v += col >= 8u ? t : 0u; }
    // (the row steps stop at row boundaries: what crosses them is the TOTAL of the rows before)
    if (W >= 32u) v += dpp_mov<kDppBcast15, 0xA, 0xF, false>(v);      // rows 1 and 3 += lane 15 of the row before
    if (W >= 64u) v += dpp_mov<kDppBcast31, 0xC, 0xF, false>(v);      // rows 2 and 3 += lane 31 (complete after the step above)
    return v;
}

// inclusive prefix sum over the 64 lanes of a wave: four row steps, then lane 15 / lane 31 broadcasts (GFX9 DPP)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += dpp_mov<kDppRowShr + 1>(v);
    v += dpp_mov<kDppRowShr + 2>(v);
    v += dpp_mov<kDppRowShr + 4>(v);
    v += dpp_mov<kDppRowShr + 8>(v);
    v += dpp_mov<kDppBcast15, 0xA, 0xF, false>(v);      // rows 1 and 3 += the total of the row before
    v += dpp_mov<kDppBcast31, 0xC, 0xF, false>(v);      // rows 2 and 3 += the total of the first half
    return v;
}

// Exact top-k of ONE tile, straight from the bit-sliced counters (run_topk without score rows): the k best
// documents of a tile under (score desc, document asc) are a superset of the tile's share of the query's k
// best (counts_to_result's partial_sort, classic_search.cpp:134-145), so K3 only has to merge tiles x k
// candidates instead of reading a score row per query twice.  Radix descent over the planes, highest first,
// on the candidate masks M (128 documents per lane): c = documents of M with the plane's bit set, summed over
// the tile with a butterfly; c >= k_rem -> the k_rem best all have the bit: M &= plane; else those c
// documents are in for sure (R |= ...), k_rem -= c, M &= ~plane.  What is left in M ties at the cut score:
// the first k_rem in document order (lane = chunk order, then word, then bit) join R.  All in registers of
// wave 0, NP rounds of ~12 VALU + log2(W) cross-lane adds; the LUT expansion and the score stores are not run.
template <int NP, bool MQ>
__device__ __forceinline__ void tile_topk(const ScanArgs& a, const uint32_t (&pl)[4][NP], uint32_t lane, uint32_t W,
                                          uint32_t tile, uint32_t qi, const uint32_t* tmeta, const uint32_t* tthr) {
    const uint32_t G = 64u / W;
    const uint32_t col = lane & (W - 1u), grp = lane / W;
    const uint32_t qraw = MQ ? qi * G + grp : qi;
    const bool live = (MQ || lane < W) && qraw < a.nq;            // lanes that hold the final planes of a real query
    const uint32_t q = qraw < a.nq ? qraw : a.nq - 1u;
    const uint32_t doc0 = tmeta[col * 3 + 2];
    uint32_t nvalid = live ? tmeta[col * 3 + 1] * 8u : 0u;         // row bytes inside the page ...
    nvalid = min(nvalid, a.num_docs > doc0 ? a.num_docs - doc0 : 0u);   // ... that belong to real documents
    const uint32_t thr = a.thresholds ? tthr[MQ ? grp : 0u] : 0u;
    uint32_t M[4], R[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t nw_ = nvalid > 32u * w ? min(nvalid - 32u * w, 32u) : 0u;
        uint32_t v = nw_ >= 32u ? 0xFFFFFFFFu : (1u << nw_) - 1u;
        if (a.thresholds) {      // count >= threshold on the planes (as the hits-only epilogue does)
            uint32_t ge = (NP < 32 && (thr >> (NP < 32 ? NP : 0)) != 0u) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < NP; ++k) ge = ((thr >> k) & 1u) ? (ge & pl[w][k]) : (ge | pl[w][k]);
            v &= ge;
        }
        M[w] = v;
        R[w] = 0u;
    }
    uint32_t krem = a.topk_k;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p) {
        uint32_t A[4], c = 0u;
#pragma unroll
        for (int w = 0; w < 4; ++w) { A[w] = M[w] & pl[w][p]; c += (uint32_t)__popc(A[w]); }
        c = group_allsum(c, W);
        const bool take = c >= krem;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            R[w] |= take ? 0u : A[w];
            M[w] = take ? A[w] : (M[w] & ~pl[w][p]);
        }
        krem -= take ? 0u : c;
    }
    auto group_excl = [&](uint32_t v, uint32_t* total) {
        const uint32_t incl = group_incl_scan(v, W, col);
        *total = __shfl(incl, lane | (W - 1u));
        return incl - v;
    };
    {   // ties at the cut score: the first krem of M in document order
        uint32_t cm = 0u, tot;
#pragma unroll
        for (int w = 0; w < 4; ++w) cm += (uint32_t)__popc(M[w]);
        uint32_t before = group_excl(cm, &tot);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t pc = (uint32_t)__popc(M[w]);
            const uint32_t allow = krem > before ? krem - before : 0u;
            uint32_t x = M[w];
            if (allow < pc) {
                uint32_t y = x;
                for (uint32_t i = 0; i < allow; ++i) y &= y - 1u;          // y = x without its lowest `allow` bits
                x &= ~y;
            }
            R[w] |= x;
            before += pc;
        }
    }
    uint32_t cs = 0u, total;
#pragma unroll
    for (int w = 0; w < 4; ++w) cs += (uint32_t)__popc(R[w]);
    uint32_t pos = group_excl(cs, &total);
    uint2* dst = a.cand + (uint64_t)q * a.cand_stride + (uint64_t)(a.tile_base + tile) * a.topk_k;
    if (live) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t x = R[w];
            while (x != 0u) {
                const uint32_t d = (uint32_t)__ffs((int)x) - 1u;
                x &= x - 1u;
                uint32_t score = 0u;
#pragma unroll
                for (int k = 0; k < NP; ++k) score |= ((pl[w][k] >> d) & 1u) << k;
                dst[pos++] = make_uint2(doc0 + 32u * w + d, score);
            }
        }
        // entries the tile does not fill are marked (K3 skips them); the pool needs no clearing between runs
        for (uint32_t i = total + col; i < a.topk_k; i += W) dst[i] = make_uint2(0xFFFFFFFFu, 0u);
    }
}

// MQ ("multi-query", short queries): the G = 64 / W lane groups of a wave belong to G
// DIFFERENT queries (q = qi*G + grp) instead of splitting one query's blocks.  Every lane
// group then walks all blocks of its own query (divided over the NW waves only): G times
// more trips per wave, so the load pipeline reaches its steady state even for 100-bp reads,
// and the cross-lane merge disappears.
// TK: the epilogue is tile_topk (run_topk without score rows) instead of the score / hit epilogues -- its own
// instantiations, so that the kernels that write scores keep their register budget (as a run-time branch
// it cost the 100-bp multi-query kernel its fourth wave per SIMD: 126 -> 131 VGPRs).
// (the tile_topk instantiations of up to 10 planes are held to the 128 VGPRs of four waves per SIMD, like the
// kernels whose epilogue they replace: the selection's masks would otherwise cost the multi-query ones a wave;
// so are the generic-H instantiations of one and two waves per group, whose half-block row loop fits them -- the
// four-wave one would spill in that loop and is left to the compiler: geometry.cpp does not choose it)
template <int NP, int NW, bool H1, typename OutT, bool MQ, typename IdxT, bool LDSS = false, bool TK = false>
__global__ __launch_bounds__(NW * 64, (((TK && H1) || (!H1 && sizeof(IdxT) == 4 && !LDSS && (NW <= 2 || TK))) && NP <= 10) ? 4 : 1) void scan_kernel(ScanArgs a) {
    // row loads stay temporal: non-temporal loads measured 18 % slower (they bypass the Infinity Cache)
    constexpr bool NT = false;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // merge buffers: [NW/2 (min 1)][NP][64 lanes] of uint4, then the 256-entry expansion table.
    // LDSS: the front of the buffer is the row staging ring (2 x 8 KiB per wave); the merge
    // buffers, needed only after the row loop, alias it.
    constexpr size_t kFront = scan_lds_front<NP, NW, sizeof(OutT), LDSS>();
    static_assert(!LDSS || kFront >= (size_t)(NW >= 2 ? NW / 2 : 1) * NP * 64 * sizeof(uint4), "ring holds the merge buffers");
    uint4* mbuf = reinterpret_cast<uint4*>(smem);
    // 16/32-bit scores expand their planes through a 256-entry table; 8-bit scores transpose in
    // registers (planes_to_bytes32) and have no table
    constexpr size_t kLut = sizeof(OutT) == 1 ? 0 : 256 * sizeof(uint4);
    uint4* lut = reinterpret_cast<uint4*>(smem + kFront);
    // per-chunk metadata of this tile (first local score slot, valid row bytes left in the
    // chunk, first document id) and per-lane-group thresholds: the epilogue reads them from
    // LDS instead of chasing a.pages[] / a.thresholds[] through global memory per iteration
    uint32_t* tmeta = reinterpret_cast<uint32_t*>(smem + kFront + kLut);   // [64][3]
    uint32_t* tthr = tmeta + 64 * 3;                                 // [64]
    if constexpr (sizeof(OutT) != 1 && !TK) {
        for (uint32_t v = threadIdx.x; v < 256u; v += NW * 64) {
            uint4 e;
            e.x = (v & 1u) | ((v & 2u) << 15);
            e.y = ((v >> 2) & 1u) | ((v & 8u) << 13);
            e.z = ((v >> 4) & 1u) | ((v & 32u) << 11);
            e.w = ((v >> 6) & 1u) | ((v & 128u) << 9);
            lut[v] = e;
        }
    }

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    COBS_STAMP(0);
    // A tile is W (power of two, <= 64) sixteen-byte column chunks of the chunk range
    // [chunk_begin, chunk_end).  With W < 64 one wave-load fetches G = 64 / W different
    // rows (terms): lane group g of wave w acts as "virtual wave" w*G + g with its own
    // stream of 8-term blocks.  Narrow tiles keep the slice of a small sub-index that
    // all queries of the batch hammer (rows x W*16 bytes) inside the 256 MB Infinity
    // Cache, and they fill the lanes when the whole index is narrower than a wave.
    const uint32_t W = a.tile_w;
    const uint32_t G = 64u / W;
    const uint32_t ntiles = (a.chunk_end - a.chunk_begin + W - 1u) / W;
    (void)ntiles;
    // tile-major order: co-resident groups read the same sub-index columns (query-major: 13 % slower)
    const uint32_t nqg = MQ ? (a.nq + G - 1u) / G : a.nq;         // query groups per tile
    const uint32_t tile = blockIdx.x / nqg;
    const uint32_t qi = blockIdx.x - tile * nqg;
    const uint32_t grp = lane / W, col = lane & (W - 1u);
    // MQ: per-lane query (groups past the end of the batch idle on the padding block)
    const uint32_t qraw = MQ ? qi * G + grp : qi;
    const bool qlive = qraw < a.nq;
    const uint32_t q = qlive ? qraw : a.nq - 1u;
    if (a.thresholds && wave == 0u && col == 0u) tthr[MQ ? grp : 0u] = a.thresholds[q];
    const uint32_t g = a.chunk_begin + tile * W + col;
    const uint32_t gc = g < a.chunk_end ? g : a.chunk_end - 1u;     // dead lanes duplicate a live one
    const uint32_t pg = gc / a.cpp;
    const uint32_t ch = gc - pg * a.cpp;
    const PageDev pdl = a.pages[pg];
    const uint8_t* lane_base = a.blob + pdl.base + (uint64_t)ch * 16u;
    const uint32_t pitch = a.pitch;
    if (threadIdx.x < W) {        // wave 0, lane group 0: one lane per chunk of the tile
        const uint32_t vb = (g < a.chunk_end && pdl.valid_bytes > ch * 16u) ? min(pdl.valid_bytes - ch * 16u, 16u) : 0u;
        tmeta[col * 3 + 0] = pdl.slot0 + ch * 128u;
        tmeta[col * 3 + 1] = vb;
        tmeta[col * 3 + 2] = pdl.doc0 + ch * 128u;
    }

    const uint64_t b0 = a.blk_off[q];
    const uint32_t nblk_q = (uint32_t)(a.blk_off[q + 1] - b0);   // table stride of query q
    const uint32_t nblk = qlive ? nblk_q : 0u;
    const uint32_t H = H1 ? 1u : a.num_hashes;
    // row indices of this lane's sub-index: [nblk + 1 blocks][hash][8]; block nblk is all padding
    const IdxT* tab = reinterpret_cast<const IdxT*>(a.table) +
                      ((b0 + q) * a.table_npages + (uint64_t)pdl.tpage * (nblk_q + 1u)) * (8ull * H);
    const uint32_t vw = MQ ? wave : wave * G + grp;  // virtual wave of this lane
    const uint32_t NV = MQ ? (uint32_t)NW : NW * G;
    // block of this lane in trip i: vw + i * NV, or the padding block when it has run out
    auto blk_of = [&](uint32_t i) -> uint64_t {
        const uint32_t bidx = vw + i * NV;
        return (uint64_t)(bidx < nblk ? bidx : nblk_q) * 8u * H;
    };

    uint32_t pl[4][NP];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int k = 0; k < NP; ++k) pl[w][k] = 0u;

    // trips of this wave (wave-uniform): as many as its first lane group needs
    uint32_t nw;
    if constexpr (MQ) {
        // the longest query among the wave's lane groups sets the trip count
        uint32_t need = nblk > wave ? (nblk - wave + NW - 1u) / NW : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) need = max(need, (uint32_t)__shfl_xor(need, off));
        nw = __builtin_amdgcn_readfirstlane(need);
    } else {
        const uint32_t first = wave * G;
        nw = nblk > first ? (nblk - first + NV - 1u) / NV : 0u;
    }
    uint32_t ea[4], eb[4];
    COBS_STAMP(1);                 // page / block-offset loads done, LUT written
    if constexpr (LDSS) {
        static_assert(!LDSS || (H1 && !MQ && sizeof(IdxT) == 4), "LDS-staged variant: H = 1, one query per group, 32-bit indices");
        const uint32_t* tab32 = reinterpret_cast<const uint32_t*>(tab);
        // LDS byte address of this wave's ring (wave-uniform: it goes into M0)
        const uint32_t ring = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem + wave * 16384u);
        const uint8_t* my = smem + wave * 16384u + lane * 16u;          // this lane's 16 bytes of row 0, buffer 0
        auto read_rows = [&](uint4 (&X)[8], uint32_t buf) {
#pragma unroll
            for (int r = 0; r < 8; ++r) X[r] = *reinterpret_cast<const uint4*>(my + buf * 8192u + r * 1024u);
        };
        if (nw > 0) {
            // VMEM queue, oldest first, at the top of step i: idx(i+1) [2 ops] | rows(i) [8 ops].
            // A step loads idx(i+2), waits for idx(i+1), issues rows(i+1) into the other buffer, and
            // only then waits for rows(i): two trips (16 KiB per wave) are in flight at every wait.
            u32x4 pa, pb, qa, qb;              // two index register sets used alternately (no copies)
            uint4 X[8];
            asm_load_idx(pa, pb, tab32 + blk_of(0));
            asm_load_idx(qa, qb, tab32 + blk_of(1));
            COBS_WAIT_VM_IDX(2, pa, pb);
            glds_rows(lane_base, pitch, pa, pb, ring);
            uint32_t i = 0;
            for (; i + 2 < nw; i += 2) {
                asm_load_idx(pa, pb, tab32 + blk_of(i + 2));
                COBS_WAIT_VM_IDX(10, qa, qb);
                glds_rows(lane_base, pitch, qa, qb, ring + 8192u);
                COBS_WAIT_VM(10);
                read_rows(X, 0u);
                absorb_block<NP>(pl, X, ea);
                asm_load_idx(qa, qb, tab32 + blk_of(i + 3));
                COBS_WAIT_VM_IDX(10, pa, pb);
                glds_rows(lane_base, pitch, pa, pb, ring);
                COBS_WAIT_VM(10);
                read_rows(X, 1u);
                absorb_block<NP>(pl, X, eb);
                retire_pair<NP>(pl, ea, eb);
            }
            // queue: idx(i+1) in (qa, qb) | rows(i) in buffer 0; one or two trips left
            if (i + 1 < nw) {
                COBS_WAIT_VM_IDX(8, qa, qb);
                glds_rows(lane_base, pitch, qa, qb, ring + 8192u);
                COBS_WAIT_VM(8);
                read_rows(X, 0u);
                absorb_block<NP>(pl, X, ea);
                COBS_WAIT_VM(0);
                read_rows(X, 1u);
                absorb_block<NP>(pl, X, eb);
                retire_pair<NP>(pl, ea, eb);
            } else {
                COBS_WAIT_VM_IDX(0, qa, qb);
                read_rows(X, 0u);
                absorb_block<NP>(pl, X, ea);
                retire_single<NP>(pl, ea);
            }
        }
        __syncthreads();                   // the merge buffers alias the rings: every wave is done reading
    } else if constexpr (H1) {
        // Three-stage software pipeline, branch-free in the steady state:
        //   row indices of trip i+2 | row loads of trip i+1 | CSA of trip i
        // so that 8..16 row loads (8..16 KiB per wave) are always in flight.
        if (nw > 0) {
            uint4 XA[8], XB[8];
            Idx8<IdxT> i0 = load_idx8(tab + blk_of(0));
            COBS_STAMP(2);         // first row indices landed
            issue_rows<NT>(XA, lane_base, pitch, i0);
            // (requesting the second index set before the first rows -- one dependent round trip less at the start --
            // measured equal on 50 / 100-bp reads and C3 and 1.4 % slower on 150-bp reads, round 3: occupancy hides it)
            Idx8<IdxT> i1 = load_idx8(tab + blk_of(1));
            COBS_STAMP(3);         // first rows landed
            uint32_t i = 0;
            for (; i + 2 < nw; i += 2) {
                // XA in flight = trip i, i1 = indices of trip i+1
                i0 = load_idx8(tab + blk_of(i + 2));
                issue_rows<NT>(XB, lane_base, pitch, i1);
                absorb_block<NP>(pl, XA, ea);
                i1 = load_idx8(tab + blk_of(i + 3));
                issue_rows<NT>(XA, lane_base, pitch, i0);
                absorb_block<NP>(pl, XB, eb);
                retire_pair<NP>(pl, ea, eb);
            }
            // XA in flight = trip i; one or two trips left
            if (i + 1 < nw) {
                issue_rows<NT>(XB, lane_base, pitch, i1);
                absorb_block<NP>(pl, XA, ea);
                absorb_block<NP>(pl, XB, eb);
                retire_pair<NP>(pl, ea, eb);
            } else {
                absorb_block<NP>(pl, XA, ea);
                retire_single<NP>(pl, ea);
            }
        }
    } else if constexpr (sizeof(IdxT) == 4) {
        // general H (aggregate_rows, reference classic_search.cpp:279-307: AND the H hash rows of each term, then count),
        // 32-bit row indices -- in HALF blocks of four terms (round 6).  Whole blocks kept three 8-row register sets live
        // (two in flight + the AND accumulator: 96 VGPRs beside 40 of planes), 189-215 VGPRs, two waves per SIMD; with
        // four-row sets (48 VGPRs) these instantiations fit the 128 VGPRs of FOUR waves per SIMD like the H = 1 kernels:
        // the same row bytes in flight per SIMD, twice the waves to hide a gather's latency behind.  The (block, half,
        // hash) triples of a wave form one stream of sub-trips through the usual three-stage pipeline -- row indices of
        // sub-trip s+2 | row loads of s+1 | AND / CSA of s; a block's first half folds into planes 0..1 and leaves its
        // fours pending, the second half adds both fours into plane 2 and ripples the eights.
        if (nw > 0) {
            uint4 XA[4], XB[4], ACC[4];
            uint32_t f4a[4] = {0u, 0u, 0u, 0u};
            const uint32_t total = nw * H * 2u;        // (even: the pipeline below always ends on a pair)
            uint32_t li = 0, lh = 0, lj = 0;           // (trip, half, hash) of the next index load
            auto next_idx = [&]() -> uint4 {           // trips >= nw point at the padding block
                const uint4 r = *reinterpret_cast<const uint4*>(tab + blk_of(li) + 8u * lj + 4u * lh);
                if (++lj == H) { lj = 0; if (++lh == 2u) { lh = 0; ++li; } }
                return r;
            };
            uint32_t cj = 0, ch = 0;                   // hash / half of the sub-trip being consumed
            auto consume = [&](const uint4 (&X)[4]) {
                if (cj == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) ACC[t] = X[t];
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { ACC[t].x &= X[t].x; ACC[t].y &= X[t].y; ACC[t].z &= X[t].z; ACC[t].w &= X[t].w; }
                }
                if (++cj == H) {
                    cj = 0;
                    const uint32_t g0 = absorb4<NP>(pl[0], ACC[0].x, ACC[1].x, ACC[2].x, ACC[3].x);
                    const uint32_t g1 = absorb4<NP>(pl[1], ACC[0].y, ACC[1].y, ACC[2].y, ACC[3].y);
                    const uint32_t g2 = absorb4<NP>(pl[2], ACC[0].z, ACC[1].z, ACC[2].z, ACC[3].z);
                    const uint32_t g3 = absorb4<NP>(pl[3], ACC[0].w, ACC[1].w, ACC[2].w, ACC[3].w);
                    if (ch == 0u) {
                        f4a[0] = g0; f4a[1] = g1; f4a[2] = g2; f4a[3] = g3;
                        ch = 1u;
                    } else {
                        const uint32_t g[4] = {g0, g1, g2, g3};
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            uint32_t e8;
                            csa(e8, pl[w][2], pl[w][2], f4a[w], g[w]);
                            ripple<NP, 3>(pl[w], e8);
                        }
                        ch = 0u;
                    }
                }
            };
            uint4 i0 = next_idx();
            issue_rows4<NT>(XA, lane_base, pitch, i0);
            uint4 i1 = next_idx();
            uint32_t sidx = 0;
            for (; sidx + 2 < total; sidx += 2) {
                i0 = next_idx();
                issue_rows4<NT>(XB, lane_base, pitch, i1);
                consume(XA);
                i1 = next_idx();
                issue_rows4<NT>(XA, lane_base, pitch, i0);
                consume(XB);
            }
            issue_rows4<NT>(XB, lane_base, pitch, i1);
            consume(XA);
            consume(XB);
        }
    } else {
        // general H, 64-bit row indices (a sub-index of 2^32 - 1 rows or more): whole blocks.
        // The (block, hash) pairs of a wave form one stream of "sub-trips" that runs through the
        // same three-stage pipeline as the H = 1 loop -- row indices of sub-trip s+2 | row loads of
        // s+1 | AND / CSA of s -- so that 8..16 rows are always in flight (the first version
        // issued a block's rows only after the previous block had been counted).
        if (nw > 0) {
            uint4 XA[8], XB[8], ACC[8];
            const uint32_t total = nw * H;
            uint32_t li = 0, lj = 0;                   // (trip, hash) of the next index load
            auto next_idx = [&]() {
                const Idx8<IdxT> r = load_idx8(tab + blk_of(li) + 8u * lj);     // trips >= nw point at the padding block
                if (++lj == H) { lj = 0; ++li; }
                return r;
            };
            uint32_t cj = 0;                           // hash of the sub-trip being consumed
            auto consume = [&](const uint4 (&X)[8]) {
                if (cj == 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) ACC[t] = X[t];
                } else {
                    and_rows(ACC, X);
                }
                if (++cj == H) {
                    cj = 0;
                    absorb_block<NP>(pl, ACC, ea);
                    retire_single<NP>(pl, ea);
                }
            };
            Idx8<IdxT> i0 = next_idx();
            issue_rows<NT>(XA, lane_base, pitch, i0);
            Idx8<IdxT> i1 = next_idx();
            uint32_t sidx = 0;
            for (; sidx + 2 < total; sidx += 2) {
                i0 = next_idx();
                issue_rows<NT>(XB, lane_base, pitch, i1);
                consume(XA);
                i1 = next_idx();
                issue_rows<NT>(XA, lane_base, pitch, i0);
                consume(XB);
            }
            if (sidx + 1 < total) {
                issue_rows<NT>(XB, lane_base, pitch, i1);
                consume(XA);
                consume(XB);
            } else {
                consume(XA);
            }
        }
    }

    COBS_STAMP(4);                 // row loop done
    // ---- merge the G lane groups of this wave (bit-sliced adds across lanes)
    for (uint32_t s = MQ ? 64u : W; s < 64u; s <<= 1) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t carry = 0u;
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) {
                const uint32_t o = __shfl_down(pl[w][kk], s);
                uint32_t h;
                csa(h, pl[w][kk], pl[w][kk], o, carry);
                carry = h;
            }
        }
    }

    // ---- merge the NW partial counters (tree over waves, bit-sliced adds) ----
#pragma unroll
    for (int s = 1; s < NW; s <<= 1) {
        uint4* buf = mbuf + (size_t)(wave / (2 * s)) * NP * 64;
        if ((wave & (2 * s - 1)) == (uint32_t)s) {
#pragma unroll
            for (int k = 0; k < NP; ++k)
                buf[k * 64 + lane] = make_uint4(pl[0][k], pl[1][k], pl[2][k], pl[3][k]);
        }
        __syncthreads();
        if ((wave & (2 * s - 1)) == 0u) {
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const uint4 o = buf[k * 64 + lane];
                uint32_t h;
                csa(h, pl[0][k], pl[0][k], o.x, c0); c0 = h;
                csa(h, pl[1][k], pl[1][k], o.y, c1); c1 = h;
                csa(h, pl[2][k], pl[2][k], o.z, c2); c2 = h;
                csa(h, pl[3][k], pl[3][k], o.w, c3); c3 = h;
            }
        }
        __syncthreads();
    }
    if constexpr (TK) {
        // run_topk without score rows: wave 0 holds the final planes and selects the tile's k best from them
        __syncthreads();                 // tile metadata / thresholds written at the start are read below (NW = 1: no barrier so far)
        // (the lane number is taken afresh from the hardware: keeping threadIdx-derived values alive across the row loop
        // is what pushed the 8-plane multi-query instantiation over its 128 registers -- 3 spills, the library's only
        // scratch user in round 3)
        if (wave == 0u)
            tile_topk<NP, MQ>(a, pl, __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), W, tile, qi, tmeta, tthr);
        return;
    }
    if constexpr (sizeof(OutT) == 1) {
        // ---- 8-bit scores with the score rows wanted: wave 0 holds the final planes; its lanes
        // transpose them to bytes in registers, stage the 128 bytes of their chunk in LDS, and all
        // threads of the group copy the tile out in 16-byte pieces (coalesced stores)
        if (a.write_counts || !a.thresholds) {
            uint8_t* stage = smem;                       // the merge buffers are done with
            if (wave == 0u && (MQ || lane < W)) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t o[8];
                    planes_to_bytes32<NP>(pl[w], o);
                    uint4* dst = reinterpret_cast<uint4*>(stage + lane * kStageStride + w * 32);
                    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
                }
            }
            __syncthreads();
            COBS_STAMP(5);             // scores staged
            const uint32_t npieces = (MQ ? 64u : W) * 8u;            // 16 documents each
#pragma unroll 1
            for (uint32_t pc0 = wave * 64u; pc0 < npieces; pc0 += NW * 64) {   // wave-uniform bounds
                const uint32_t pc = pc0 + lane;
                const bool act = pc < npieces;
                const uint32_t L = act ? pc >> 3 : 0u, rb = (pc & 7u) * 2u;   // lane that staged it, first row byte
                const uint32_t chunk = MQ ? (L & (W - 1u)) : L;
                const uint32_t q2raw = MQ ? qi * G + L / W : qi;
                const uint32_t q2 = q2raw < a.nq ? q2raw : a.nq - 1u;
                const uint32_t vb = tmeta[chunk * 3 + 1];                     // valid row bytes of the chunk
                const bool valid = act && q2raw < a.nq && rb < vb;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + L * kStageStride + rb * 8u);
                const uint32_t slot = tmeta[chunk * 3 + 0] + rb * 8u;
                if (valid && a.write_counts) {
                    OutT* crow = reinterpret_cast<OutT*>(a.counts) + (uint64_t)q2 * a.counts_stride + a.counts_offset;
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    // (two 8-byte stores per piece.  Round 4 measured the alternatives, interleaved inside one process: one 16-byte
                    // store where all slots are multiples of 16 is 2.8 % SLOWER on 50-bp reads and 1.3 % on 150-bp reads; 8-byte
                    // pieces laid out so that a wave-store covers 512 contiguous bytes came out 4 % faster in one build and 2 %
                    // slower in the next -- below what separates two processes on one box: profiles/r04_short_read_experiments.txt)
                    u32x2* dst = reinterpret_cast<u32x2*>(crow + slot);       // slots are multiples of 8, not of 16
                    const u32x2 v0 = {v.x, v.y}, v1 = {v.z, v.w};
                    dst[0] = v0;
                    if (rb + 1u < vb) dst[1] = v1;
                }
                if (a.thresholds) {
                    // counts_to_result filter: score >= threshold over real documents only
                    const uint32_t thr = tthr[MQ ? L / W : 0u];
                    const uint32_t doc = tmeta[chunk * 3 + 2] + rb * 8u;
                    const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
                    uint32_t mask = 0u;
                    if (valid) {
                        const uint32_t nd = rb + 1u < vb ? 16u : 8u;
#pragma unroll
                        for (uint32_t d = 0; d < 16; ++d) {
                            const uint32_t cnt = (vv[d >> 2] >> ((d & 3u) * 8u)) & 0xFFu;
                            if (d < nd && cnt >= thr && doc + d < a.num_docs) mask |= 1u << d;
                        }
                    }
                    if (__any(mask != 0u)) {
                        const uint32_t n = __popc(mask);
                        const uint32_t incl = wave_incl_scan_dpp(n);
                        const uint32_t total = __shfl(incl, 63);
                        unsigned long long base = 0ull;
                        if (lane == 63u) base = atomicAdd(a.hit_count, (unsigned long long)total);
                        base = __shfl(base, 63);
                        unsigned long long pos = base + incl - n;
                        while (mask != 0u) {
                            const uint32_t d = (uint32_t)__ffs((int)mask) - 1u;
                            mask &= mask - 1u;
                            if (pos < a.hit_cap)
                                a.hits[pos] = HitDev{q2, a.part, doc + d, (vv[d >> 2] >> ((d & 3u) * 8u)) & 0xFFu};
                            ++pos;
                        }
                    }
                }
            }
            COBS_STAMP(6);             // scores stored
            return;
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < NP; ++k)
            mbuf[k * 64 + lane] = make_uint4(pl[0][k], pl[1][k], pl[2][k], pl[3][k]);
    }
    __syncthreads();

    COBS_STAMP(5);                 // merged planes are in LDS
    // ---- expand planes -> per-document counts; every thread handles row bytes ----
    const uint32_t* planes = reinterpret_cast<const uint32_t*>(mbuf);   // [NP][64*4 words]
    if (!a.write_counts && a.thresholds) {
        // Hits only (threshold > 0, the scores themselves are not wanted): compare in bit-sliced
        // form -- one 32-bit word = 32 documents, count >= threshold in NP boolean ops -- and
        // expand nothing unless a document passes.  With the usual thresholds hardly any word
        // holds a hit, so this replaces the whole LUT expansion (short reads: a third of the kernel).
        const uint32_t nwords = MQ ? 256u : W * 4u;
#pragma unroll 1
        for (uint32_t w0 = wave * 64u; w0 < nwords; w0 += NW * 64) {     // wave-uniform bounds
            const uint32_t w = w0 + lane;
            const bool act = w < nwords;
            const uint32_t pl_lane = act ? w >> 2 : 0u, comp = w & 3u;
            const uint32_t chunk = MQ ? (pl_lane & (W - 1u)) : pl_lane;
            const uint32_t q2raw = MQ ? qi * G + pl_lane / W : qi;
            const uint32_t q2 = q2raw < a.nq ? q2raw : a.nq - 1u;
            const uint32_t thr = tthr[MQ ? pl_lane / W : 0u];
            const uint32_t vbc = tmeta[chunk * 3 + 1];                       // valid row bytes of the chunk (0: dead)
            const bool valid = act && vbc != 0u && q2raw < a.nq;
            // ge bit d = (count of document d >= thr), from the lowest plane up:
            //   threshold bit 1: ge &= plane, threshold bit 0: ge |= plane
            uint32_t ge = (NP < 32 && (thr >> (NP < 32 ? NP : 0)) != 0u) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const uint32_t pk = planes[(k * 64 + pl_lane) * 4 + comp];
                ge = ((thr >> k) & 1u) ? (ge & pk) : (ge | pk);
            }
            // real documents only: row bytes inside the page, document ids below num_docs
            const uint32_t vb = vbc > comp * 4u ? vbc - comp * 4u : 0u;                       // valid bytes of the word
            if (vb < 4u) ge &= vb == 0u ? 0u : (1u << (vb * 8u)) - 1u;
            const uint32_t doc0 = tmeta[chunk * 3 + 2] + comp * 32u;
            const uint32_t nd = a.num_docs > doc0 ? a.num_docs - doc0 : 0u;
            if (nd < 32u) ge &= nd == 0u ? 0u : (1u << nd) - 1u;
            if (!valid) ge = 0u;
            if (__any(ge != 0u)) {
                const uint32_t n = __popc(ge);
                const uint32_t incl = wave_incl_scan_dpp(n);
                const uint32_t total = __shfl(incl, 63);
                unsigned long long base = 0ull;
                if (lane == 63u) base = atomicAdd(a.hit_count, (unsigned long long)total);
                base = __shfl(base, 63);
                unsigned long long pos = base + incl - n;
                while (ge != 0u) {
                    const uint32_t d = (uint32_t)__ffs((int)ge) - 1u;
                    ge &= ge - 1u;
                    uint32_t score = 0u;
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                        score |= ((planes[(k * 64 + pl_lane) * 4 + comp] >> d) & 1u) << k;
                    if (pos < a.hit_cap) a.hits[pos] = HitDev{q2, a.part, doc0 + d, score};
                    ++pos;
                }
            }
        }
        return;
    }
    if constexpr (sizeof(OutT) == 1) return;         // 8-bit scores left through one of the two paths above
    const uint32_t nbytes = MQ ? 1024u : W * 16u;    // MQ: every lane group holds a query's tile
    // MQ: 1024 / (NW * 64) = 4..16 iterations per thread.  Unrolled by four so that the score stores
    // of consecutive iterations use different registers: with one register set the next iteration's
    // first write to the store's data registers waits (vmcnt) until the store has left the CU --
    // measured 2 500 cycles per iteration, half of a short-read work-group's life.
    // (A counted loop: with the thread-dependent start as induction variable the unroller gives up.)
    constexpr int kUnroll = MQ ? 4 : 1;
    const uint32_t niter = MQ ? 1024u / (NW * 64) : (nbytes + NW * 64 - 1u) / (NW * 64);
#pragma unroll kUnroll
    for (uint32_t it = 0; it < niter; ++it) {
        const uint32_t b = threadIdx.x + it * (NW * 64);             // row byte inside the tile
        if (!MQ && b >= nbytes) break;
        const uint32_t pl_lane = b >> 4, cb = b & 15u;               // lane that held the planes
        const uint32_t chunk = MQ ? (pl_lane & (W - 1u)) : pl_lane;
        const uint32_t q2raw = MQ ? qi * G + pl_lane / W : qi;
        const uint32_t q2 = q2raw < a.nq ? q2raw : a.nq - 1u;
        const uint32_t thr = a.thresholds ? tthr[MQ ? pl_lane / W : 0u] : 0u;
        OutT* crow = reinterpret_cast<OutT*>(a.counts) + (uint64_t)q2 * a.counts_stride + a.counts_offset;
        const bool valid = q2raw < a.nq && cb < tmeta[chunk * 3 + 1];

        // one LDS lookup spreads the 8 document bits of a plane byte into 8 sixteen-bit
        // fields (4 dwords); shifting the dwords by the plane number adds that plane to all
        // 8 counters at once.  Planes 16.. go to a second accumulator (32-bit scores).
        uint32_t lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};
        const uint32_t sh = (cb & 3u) * 8u;
        uint32_t cnt[8];
        if constexpr (sizeof(OutT) == 1) {
            static_assert(sizeof(OutT) != 1 || NP <= 8, "8-bit scores hold at most 8 planes");
            const uint2* lut8 = reinterpret_cast<const uint2*>(lut);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const uint32_t v = (planes[(k * 64 + pl_lane) * 4 + (cb >> 2)] >> sh) & 0xFFu;
                const uint2 e = lut8[v];
                lo[0] |= e.x << k; lo[1] |= e.y << k;
            }
#pragma unroll
            for (int d = 0; d < 8; ++d) cnt[d] = (lo[d >> 2] >> ((d & 3) * 8)) & 0xFFu;
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const uint32_t v = (planes[(k * 64 + pl_lane) * 4 + (cb >> 2)] >> sh) & 0xFFu;
                const uint4 e = lut[v];
                if (k < 16) {
                    lo[0] |= e.x << k; lo[1] |= e.y << k; lo[2] |= e.z << k; lo[3] |= e.w << k;
                } else {
                    hi[0] |= e.x << (k - 16); hi[1] |= e.y << (k - 16);
                    hi[2] |= e.z << (k - 16); hi[3] |= e.w << (k - 16);
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                cnt[2 * m] = (lo[m] & 0xFFFFu) | ((hi[m] & 0xFFFFu) << 16);
                cnt[2 * m + 1] = (lo[m] >> 16) | (hi[m] & 0xFFFF0000u);
            }
        }
        const uint32_t slot = tmeta[chunk * 3 + 0] + cb * 8u;
        if (valid && a.write_counts) {
            if constexpr (sizeof(OutT) == 1) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                u32x2* dst = reinterpret_cast<u32x2*>(crow + slot);
                const u32x2 v = {lo[0], lo[1]};
                *dst = v;          // (non-temporal stores measured equal: scripts/ab.py, 50..150-bp reads, C2, C3)
            } else if constexpr (sizeof(OutT) == 2) {
                u32x4* dst = reinterpret_cast<u32x4*>(crow + slot);
                const u32x4 v = {lo[0], lo[1], lo[2], lo[3]};
                *dst = v;          // (non-temporal stores measured equal: scripts/ab.py, 50..150-bp reads, C2, C3)
            } else {
                *reinterpret_cast<uint4*>(crow + slot) = make_uint4(cnt[0], cnt[1], cnt[2], cnt[3]);
                *reinterpret_cast<uint4*>(crow + slot + 4) = make_uint4(cnt[4], cnt[5], cnt[6], cnt[7]);
            }
        }
        if (a.thresholds) {
            // counts_to_result filter: score >= threshold over real documents only
            const uint32_t doc = tmeta[chunk * 3 + 2] + cb * 8u;
            uint32_t mask = 0u;
            if (valid) {
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    if (cnt[d] >= thr && doc + d < a.num_docs) mask |= 1u << d;
            }
            if (__any(mask != 0u)) {
                const uint32_t n = __popc(mask);
                const uint32_t incl = wave_incl_scan_dpp(n);
                const uint32_t total = __shfl(incl, 63);
                unsigned long long base = 0ull;
                if (lane == 63u) base = atomicAdd(a.hit_count, (unsigned long long)total);
                base = __shfl(base, 63);
                unsigned long long pos = base + incl - n;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    if (mask & (1u << d)) {
                        if (pos < a.hit_cap) a.hits[pos] = HitDev{q2, a.part, doc + d, cnt[d]};
                        ++pos;
                    }
                }
            }
        }
    }
    COBS_STAMP(6);                 // scores stored
}

// ---------------------------------------------------------------------------
// K3: exact top-k selection per query (the device side of counts_to_result's
// partial_sort, reference classic_search.cpp:127-145, for all three Score widths of
// :453-504): find the score s* of the k-th best document with a radix descent over the
// score bits (histogram levels of at most 12 bits each: one level for 8/10/12-bit scores,
// two up to 24 bits, three for 32-bit scores), emit every document with score > s* and, in
// ascending document order, as many documents with score == s* as are still needed, then
// order the <= k survivors by (score desc, document asc) in LDS (bitonic sort on
// (~score, doc) keys) -- the host copies the result as is.
// One work-group (4 waves) per query; wave w owns the contiguous quarter w of the
// documents so that ballot prefixes keep document order.

__device__ __forceinline__ uint32_t wave_prefix(unsigned long long mask, uint32_t lane) {
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t* total) {
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += t;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

// eight consecutive scores (u8, u16 or u32) starting at document i (a multiple of 8)
template <typename ST>
__device__ __forceinline__ void load_scores8(const ST* row, uint32_t i, uint32_t (&s)[8]) {
    if constexpr (sizeof(ST) == 1) {
        const uint2 v = *reinterpret_cast<const uint2*>(row + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ((j < 4 ? v.x : v.y) >> ((j & 3) * 8)) & 0xFFu;
    } else if constexpr (sizeof(ST) == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
    } else {
        const uint4 a = *reinterpret_cast<const uint4*>(row + i);
        const uint4 b = *reinterpret_cast<const uint4*>(row + i + 4);
        s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
        s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    }
}

// Dynamic LDS: hist[4 waves][NB] (NB = 2^level_bits <= 4096; reused by every level and, after
// the emission, as the sort buffer: 8192 eight-byte keys) | partial[256] | sh[16]
// POOL: the input is not a score row but the candidate pool of run_topk without score rows (K2's tile_topk):
// nslots (document, score) entries per query in ascending document order, unused ones marked with
// document 0xFFFFFFFF; the same selection and ordering over tiles x k candidates.
template <typename ST, bool POOL>
__device__ __forceinline__ void load_elems8(const void* rowp, uint32_t i, uint32_t w1, uint32_t doc_base, uint32_t thr,
                                            uint32_t (&s)[8], uint32_t (&d)[8], uint32_t& okmask) {
    okmask = 0u;
    if constexpr (POOL) {
        const uint4* e = reinterpret_cast<const uint4*>(reinterpret_cast<const uint2*>(rowp) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 v = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
            if (i + 2 * j < w1) v = e[j];            // two entries per load; the row is padded to 8 entries, i is a multiple of 8
            d[2 * j] = v.x; s[2 * j] = v.y; d[2 * j + 1] = v.z; s[2 * j + 1] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i + j < w1 && d[j] != 0xFFFFFFFFu && s[j] >= thr) okmask |= 1u << j;
    } else {
        load_scores8<ST>(reinterpret_cast<const ST*>(rowp), i, s);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            d[j] = doc_base + i + j;
            if (i + j < w1 && s[j] >= thr) okmask |= 1u << j;
        }
    }
}

template <typename ST, bool POOL = false>
__global__ __launch_bounds__(256) void topk_kernel(TopkArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t NB = 1u << a.level_bits;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    uint32_t* partial = hist + 4u * NB;
    uint32_t* sh = partial + 256;
    const uint32_t q = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const void* row = POOL ? (const void*)(reinterpret_cast<const uint2*>(a.counts) + (uint64_t)q * a.counts_stride)
                           : (const void*)(reinterpret_cast<const ST*>(a.counts) + (uint64_t)q * a.counts_stride + a.counts_offset);
    const uint32_t thr = a.thresholds ? a.thresholds[q] : 0u;
    uint32_t n = a.nslots;                       // real documents among the local slots (POOL: pool entries)
    if constexpr (!POOL) {
        if (a.doc_base >= a.num_docs) n = 0;
        else if (a.num_docs - a.doc_base < n) n = a.num_docs - a.doc_base;
    }
    const uint32_t k = a.k;
    // wave w owns the contiguous document range [w0, w1); 512 documents per iteration
    const uint32_t per = ((n + 3u) / 4u + 511u) / 512u * 512u;
    const uint32_t w0 = wave * per < n ? wave * per : n;
    const uint32_t w1 = w0 + per < n ? w0 + per : n;

    // ---- radix descent: after level l the top (l+1)*level_bits bits of s* are known
    uint32_t prefix = 0, n_above = 0, take_all = 0;
    uint32_t bits_left = a.score_bits;           // bits below the known prefix
    uint32_t* myh = hist + wave * NB;
    for (uint32_t level = 0; level < a.levels; ++level) {
        const uint32_t lb = bits_left < a.level_bits ? bits_left : a.level_bits;     // bits of this level
        const uint32_t shift = bits_left - lb;
        const uint32_t nb = 1u << lb, mask = nb - 1u;
        for (uint32_t i = tid; i < 4u * NB; i += 256) hist[i] = 0;
        __syncthreads();
        for (uint32_t i0 = w0; i0 < w1; i0 += 512) {
            const uint32_t i = i0 + lane * 8u;
            if (i < w1) {
                uint32_t sc[8], dc[8], ok;
                load_elems8<ST, POOL>(row, i, w1, a.doc_base, thr, sc, dc, ok);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t s = sc[j];
                    // the bits above this level must equal the prefix found so far
                    const bool in = level == 0 || (bits_left >= 32u ? true : (s >> bits_left) == prefix);
                    if ((ok >> j & 1u) && in) atomicAdd(&myh[(s >> shift) & mask], 1u);
                }
            }
        }
        __syncthreads();
        {   // parallel search of the bin holding the k-th best: per-thread segment sums, then thread 0
            const uint32_t seg = (nb + 255u) / 256u;
            uint32_t sum = 0;
            for (uint32_t b = tid * seg; b < (tid + 1) * seg && b < nb; ++b)
                sum += hist[b] + hist[NB + b] + hist[2 * NB + b] + hist[3 * NB + b];
            partial[tid] = sum;
            __syncthreads();
            if (tid == 0) {
                uint32_t above = n_above;
                int t = 255;
                for (; t >= 0; --t) {
                    if (above + partial[t] >= k) break;
                    above += partial[t];
                }
                int hb = -1;
                if (t >= 0) {
                    int b = (int)((uint32_t)(t + 1) * seg) - 1;
                    if (b >= (int)nb) b = (int)nb - 1;
                    for (; b >= (int)((uint32_t)t * seg); --b) {
                        const uint32_t c = hist[b] + hist[NB + b] + hist[2 * NB + b] + hist[3 * NB + b];
                        if (above + c >= k) { hb = b; break; }
                        above += c;
                    }
                }
                sh[0] = hb < 0 ? 0u : (uint32_t)hb;
                sh[1] = above;
                sh[2] = hb < 0 ? 1u : 0u;        // fewer than k passing documents: take them all
            }
            __syncthreads();
        }
        const uint32_t hb = sh[0];
        n_above = sh[1];
        take_all = sh[2];
        prefix = (lb >= 32u ? 0u : (prefix << lb)) | hb;
        bits_left = shift;
        if (take_all) break;                     // only possible at level 0 (block-uniform)
        if (level + 1 < a.levels) __syncthreads();        // hist is zeroed again
    }
    uint32_t cut = take_all ? thr : prefix;
    // ties: documents with score == cut, per wave (the last level's per-wave histogram bins)
    uint32_t eq_base = 0, eq_total = 0;
    if (!take_all) {
        const uint32_t lastbits = a.score_bits - (a.levels - 1u) * a.level_bits;
        const uint32_t bin = cut & ((lastbits >= 32u ? 0u : (1u << lastbits)) - 1u);
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t c = hist[w * NB + bin];
            if (w < wave) eq_base += c;
            eq_total += c;
        }
    }
    const uint32_t need_eq = take_all ? 0u : k - n_above;
    if (tid == 0) sh[5] = 0;                      // emission cursor of the documents above the cut
    __syncthreads();
    // ---- emission
    uint2* out = a.out + (uint64_t)q * (a.out_stride ? a.out_stride : k);
    for (uint32_t i0 = w0; i0 < w1; i0 += 512) {
        const uint32_t i = i0 + lane * 8u;
        uint32_t s8[8], d8[8];
        uint32_t gt = 0, eq = 0;
        if (i < w1) {
            uint32_t ok;
            load_elems8<ST, POOL>(row, i, w1, a.doc_base, thr, s8, d8, ok);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t s = s8[j];
                const bool pass = (ok >> j & 1u) != 0u;
                if (pass && (take_all || s > cut)) gt |= 1u << j;
                if (pass && !take_all && s == cut) eq |= 1u << j;
            }
        }
        if (__any(gt != 0u)) {
            uint32_t total;
            const uint32_t excl = wave_excl_scan((uint32_t)__popc(gt), lane, &total);
            uint32_t base = 0;
            if (lane == 63u) base = atomicAdd(&sh[5], total);
            base = __shfl(base, 63);
            uint32_t pos = base + excl;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (gt & (1u << j)) out[pos++] = make_uint2(d8[j], s8[j]);
        }
        if (__any(eq != 0u)) {
            uint32_t total;
            const uint32_t excl = wave_excl_scan((uint32_t)__popc(eq), lane, &total);
            uint32_t r = eq_base + excl;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (eq & (1u << j)) {
                    if (r < need_eq) out[n_above + r] = make_uint2(d8[j], s8[j]);
                    ++r;
                }
            eq_base += total;
        }
    }
    __syncthreads();
    const uint32_t cnt = take_all ? sh[5] : n_above + (eq_total < need_eq ? eq_total : need_eq);
    if (tid == 0 && a.out_count) a.out_count[q] = cnt;
    // (one tile of the candidate pool: what the survivors leave of its k entries is marked unused, as tile_topk does)
    if (a.pad_out)
        for (uint32_t i = cnt + tid; i < k; i += 256) out[i] = make_uint2(0xFFFFFFFFu, 0u);
    // ---- order the survivors: (score desc, doc asc) = ascending (~score << 32 | doc)
    if (a.sort_limit && cnt > 1u && cnt <= a.sort_limit) {      // block-uniform condition
        __syncthreads();                          // everybody has read sh[]: the key area may overlap it
        unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);
        uint32_t m = 2;
        while (m < cnt) m <<= 1;
        for (uint32_t i = tid; i < m; i += 256) {
            unsigned long long kv = ~0ull;        // padding sorts last
            if (i < cnt) {
                const uint2 e = out[i];
                kv = ((unsigned long long)(~e.y) << 32) | e.x;
            }
            key[i] = kv;
        }
        __syncthreads();
        for (uint32_t size = 2; size <= m; size <<= 1) {
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t t = tid; t < (m >> 1); t += 256) {
                    const uint32_t lo = (t / stride) * (stride << 1) + (t % stride), hi = lo + stride;
                    const bool up = ((lo & size) == 0u);
                    const unsigned long long x = key[lo], y = key[hi];
                    if ((x > y) == up) { key[lo] = y; key[hi] = x; }
                }
                __syncthreads();
            }
        }
        for (uint32_t i = tid; i < cnt; i += 256) {
            const unsigned long long kv = key[i];
            out[i] = make_uint2((uint32_t)kv, ~(uint32_t)(kv >> 32));
        }
    }
}

// ---------------------------------------------------------------------------
// The threshold filter over ACCUMULATED scores (reference classic_search.cpp:127-132).  A streamed sub-index that is
// larger than a stream buffer is counted row range by row range (pass.cpp): K2 sees partial counts there and cannot
// compare them with a threshold.  Its ranges add up in a scratch matrix of the sub-index's own width (no score rows of
// the whole index), and after the last range this kernel does what K2's epilogue does for a sub-index it sees whole:
// score >= threshold over real documents -> (query, file, document, score) records into the batch's hit pool, one
// wave-aggregated atomic per wave.  One thread per (query, 8 consecutive slots).
template <typename ST>
__global__ __launch_bounds__(256) void select_rows_kernel(SelectRowsArgs a) {
    const uint32_t groups = (a.nslots + 7u) / 8u;
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t qq = gid / groups;
    const bool live = qq < a.nq;                      // (no early return: the scan below runs over whole waves)
    const uint32_t q = live ? (uint32_t)qq : 0u;
    const uint32_t i = (uint32_t)(gid - qq * groups) * 8u;
    uint32_t s8[8];
    uint32_t mask = 0u;
    if (live) {
        load_scores8<ST>(reinterpret_cast<const ST*>(a.scores) + (uint64_t)q * a.stride, i, s8);
        const uint32_t thr = a.thresholds[q];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i + j < a.nslots && a.doc0 + i + j < a.num_docs && s8[j] >= thr) mask |= 1u << j;
    }
    if (__any(mask != 0u)) {
        const uint32_t n = (uint32_t)__popc(mask);
        const uint32_t incl = wave_incl_scan_dpp(n);
        const uint32_t total = __shfl(incl, 63);
        unsigned long long base = 0ull;
        if (lane == 63u) base = atomicAdd(a.hit_count, (unsigned long long)total);
        base = __shfl(base, 63);
        unsigned long long pos = base + incl - n;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (mask & (1u << j)) {
                if (pos < a.hit_cap) a.hits[pos] = HitDev{q, a.part, a.doc0 + i + (uint32_t)j, s8[j]};
                ++pos;
            }
    }
}

// ---------------------------------------------------------------------------
// Construction: the reference sets bit (doc % 8) of byte doc / 8
// of row XXH64(canon(term), seed j) % signature_size for every term of every document
// (cobs/construction/classic_index.cpp:40-73).  One thread per position of the term text; the
// text is a sequence of stretches (documents.hpp): in a raw stretch every k-gram is a term, in a
// line stretch a position starts a term if the next k characters hold no '\n'.  With
// canonicalize = 1 the reference hashes the canonicalised buffer even when it holds invalid
// characters (mapped to 0), which the generic byte view reproduces; 31-mers of valid bases --
// nearly all of a DNA collection -- take the register path of the query hash kernel instead
// (unaligned dword loads, canon31, unrolled XXH64).
__device__ __forceinline__ bool all_acgt(uint32_t w) {
    // (c >> 1) & 3 maps A C G T to 0 1 3 2; v_perm rebuilds the letters from that code
    const uint32_t code = (w >> 1) & 0x03030303u;
    return __builtin_amdgcn_perm(0u, 0x47544341u, code) == w;
}
__device__ __forceinline__ bool has_newline(uint32_t w) {
    const uint32_t x = w ^ 0x0A0A0A0Au;
    return ((x - 0x01010101u) & ~x & 0x80808080u) != 0u;
}

// Where a term's bit goes.  Scattered atomics run at 22-27 G/s on this part whatever their locality
// (they leave the L2 as 32-byte memory-side requests, profiles/r02_atomic_probe.txt) while plain
// byte stores reach 41 G/s and more: in byte-map mode a term stores a byte into its document's
// plane and pack_bytemap_kernel turns the planes of the launch into matrix bits afterwards.
__device__ __forceinline__ void set_term_bit(const BuildArgs& a, uint32_t doc, uint64_t row) {
    if (a.bytemap != nullptr) {
        a.bytemap[(uint64_t)(doc - a.col_base) * a.bm_stride + row] = 1;
    } else {
        const uint64_t byte_in_row = doc >> 3;
        const uint32_t bit = 1u << ((uint32_t)(byte_in_row & 3u) * 8u + (doc & 7u));
        atomicOr(a.matrix + (row * a.row_bytes + byte_in_row) / 4u, bit);
    }
}

// One thread per (four consecutive rows, one 32-document word of the matrix row): reads the rows'
// bytes from every plane whose column falls into the word (dword loads, coalesced across the
// rows of a wave), ORs the bits into the four words.  Launches of one build are ordered on one
// stream and every (row, word) belongs to one thread, so the read-modify-write needs no atomic.
__global__ __launch_bounds__(256) void pack_bytemap_kernel(PackArgs a) {
    const uint32_t w0 = a.col_base >> 5, w1 = (a.col_base + a.ndocs - 1u) >> 5;      // words the launch touches
    const uint64_t nquads = (a.rows + 3u) / 4u;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t quad = gid % nquads;                   // consecutive threads = consecutive rows
    const uint32_t w = w0 + (uint32_t)(gid / nquads);
    if (w > w1) return;
    const uint64_t r0 = quad * 4u;
    const uint32_t c0 = max(a.col_base, w << 5), c1 = min(a.col_base + a.ndocs, (w + 1u) << 5);
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t x = *reinterpret_cast<const uint32_t*>(a.bytemap + (uint64_t)(c - a.col_base) * a.bm_stride + r0);
        const uint32_t b = c & 31u;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] |= ((x >> (8 * i)) & 1u) << b;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (r0 + i < a.rows && acc[i] != 0u) {
            uint32_t* m = a.matrix + ((r0 + i) * a.row_bytes) / 4u + w;
            *m |= acc[i];
        }
    }
}

__global__ __launch_bounds__(256) void build_kernel(BuildArgs a, uint64_t total_bytes) {
    // the stretch of the block's first position (wave-uniform search), then a few steps per thread
    const uint64_t base = (uint64_t)blockIdx.x * 256u;
    uint32_t lo = 0, hi = a.nsegs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.seg_off[mid] <= base) lo = mid; else hi = mid;
    }
    const uint64_t gid = base + threadIdx.x;
    if (gid >= total_bytes) return;
    while (a.seg_off[lo + 1] <= gid) ++lo;            // seg_off[nsegs] = total_bytes > gid
    const uint32_t k = a.term_size;
    if (gid + k > a.seg_off[lo + 1]) return;          // the term would leave its stretch
    const uint32_t colw = a.seg_col[lo];
    if (colw == kBuildGapStretch) return;
    const bool raw = (colw & kBuildRawStretch) != 0u;
    const uint32_t doc = colw & ~kBuildRawStretch;
    const uint8_t* p = a.text + gid;
    if (k == 31u) {
        // the 31-mer and one following byte as 8 (unaligned) dwords; the text buffer is padded
        uint32_t f[8];
        {
            const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
            const uint32_t* w = reinterpret_cast<const uint32_t*>(p - mis);
            uint32_t r[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) r[j] = w[j];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                f[j] = mis == 0 ? r[j] : (uint32_t)(((uint64_t)r[j] | ((uint64_t)r[j + 1] << 32)) >> (8 * mis));
        }
        f[7] &= 0x00FFFFFFu;
        if (!raw) {
            bool nl = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) nl |= has_newline(f[j]);
            if (nl) return;                           // the term would span a sequence boundary
        }
        bool fast = true;
        uint32_t c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = f[j];
        if (a.canonicalize != 0) {
#pragma unroll
            for (int j = 0; j < 7; ++j) fast &= all_acgt(f[j]);
            fast &= all_acgt(f[7] | 0x41000000u);
            if (fast) canon31(f, c);
        }
        if (fast) {
            for (uint32_t j = 0; j < a.num_hashes; ++j) {
                const uint64_t row = fast_mod(xxh64_31(c, (uint64_t)j), a.signature_size, a.magic);
                set_term_bit(a, doc, row);
            }
            return;
        }
    } else if (!raw) {
        for (uint32_t i = 0; i < k; ++i)
            if (p[i] == '\n') return;                 // the term would span a sequence boundary
    }
    KmerView kv{p, k, 0u};
    if (a.canonicalize != 0) {
        uint32_t mode = 1;
        for (uint32_t s = 0; s < k / 2; ++s) {
            const int f = (int)fwd_base(p[s]);
            const int r = (int)rev_base(p[k - 1 - s]);
            if (f < r) break;
            if (f > r) { mode = 2; break; }
        }
        kv.mode = mode;
    }
    for (uint32_t j = 0; j < a.num_hashes; ++j) {
        const uint64_t row = fast_mod(xxh64_view(kv, (uint64_t)j), a.signature_size, a.magic);
        set_term_bit(a, doc, row);
    }
}

// ---------------------------------------------------------------------------
// procedural index bits (same definition as the checker's generator)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// classic_construct_random (construction/classic_index.cpp:661-725): every document is
// document_size random 31-mers; each is canonicalised, hashed and its bit set.  One thread per
// (document, k-mer); the 31 bases are the low 62 bits of mix64(mix64(seed ^ doc) + j), two bits
// per base (A C G T), first base in the lowest bits.
__global__ __launch_bounds__(256) void random_build_kernel(RandomBuildArgs a, uint64_t total) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint64_t doc = gid / a.document_size, j = gid - doc * a.document_size;
    uint64_t bits = mix64(mix64(a.seed ^ (a.doc0 + doc)) + j);
    uint32_t f[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t code = (uint32_t)(bits >> (2 * (4 * w + b))) & 3u;
            // A 0x41, C 0x43, G 0x47, T 0x54
            const uint32_t ch = code == 0 ? 0x41u : code == 1 ? 0x43u : code == 2 ? 0x47u : 0x54u;
            v |= ch << (8 * b);
        }
        f[w] = v;
    }
    f[7] &= 0x00FFFFFFu;
    uint32_t c[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) c[w] = f[w];
    canon31(f, c);
    const uint64_t col = a.doc0 + doc;
    const uint64_t byte_in_row = col >> 3;
    const uint32_t bit = 1u << ((uint32_t)(byte_in_row & 3u) * 8u + (uint32_t)(col & 7u));
    for (uint32_t h = 0; h < a.num_hashes; ++h) {
        const uint64_t row = fast_mod(xxh64_31(c, (uint64_t)h), a.signature_size, a.magic);
        atomicOr(a.matrix + (row * a.row_bytes + byte_in_row) / 4u, bit);
    }
}

// classic_combine (construction/classic_index.cpp:195-327): row r of the output is the rows r of
// the inputs concatenated at BIT granularity (input i contributes its row_bits[i] documents).
// One thread per output byte.
__global__ __launch_bounds__(256) void combine_kernel(CombineArgs a) {
    const uint64_t total = a.rows * a.dst_row_bytes;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = i / a.dst_row_bytes;
        const uint64_t ob = i - row * a.dst_row_bytes;
        uint64_t bit = ob * 8;                                  // first output document of this byte
        // source holding document `bit`: last s with bit_off[s] <= bit
        uint32_t lo = 0, hi = a.nsrc;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.bit_off[mid] <= bit) lo = mid; else hi = mid;
        }
        uint32_t sidx = lo, v = 0;
#pragma unroll 1
        for (uint32_t b = 0; b < 8 && bit < a.bit_off[a.nsrc]; ++b, ++bit) {
            while (bit >= a.bit_off[sidx + 1]) ++sidx;          // sources without documents are skipped
            const uint64_t sb = bit - a.bit_off[sidx];
            const uint8_t byte = a.src[sidx][row * a.src_row_bytes[sidx] + (sb >> 3)];
            v |= ((uint32_t)(byte >> (sb & 7u)) & 1u) << b;
        }
        a.dst[i] = (uint8_t)v;
    }
}

__device__ __forceinline__ uint64_t synth_word(uint64_t seed, uint32_t page, uint64_t row, uint64_t w) {
    const uint64_t key = mix64(seed ^ mix64(((uint64_t)page << 40) ^ row));
    const uint64_t c = key + w * 6;
    const uint64_t x = mix64(c) & mix64(c + 1);
    const uint64_t y = mix64(c + 2) & mix64(c + 3) & mix64(c + 4) & mix64(c + 5);
    return x | y;
}

// grid: blockIdx.y = local page, grid-stride over (row, 8-byte word) of that page
__global__ __launch_bounds__(256) void synth_kernel(SynthArgs a) {
    const uint32_t p = blockIdx.y;
    const PageDev pd = a.pages[p];
    const uint32_t wpr = a.pitch / 8u;                    // words per HBM row
    const uint64_t nwords = (pd.sig + 1) * (uint64_t)wpr; // incl. the zero row
    const uint32_t fpage = a.first_page + p;
    const uint64_t first_doc = (uint64_t)fpage * a.page_docs;
    const uint64_t live = a.num_docs > first_doc ? a.num_docs - first_doc : 0;   // real documents of the page
    uint64_t* dst = reinterpret_cast<uint64_t*>(a.blob + pd.base);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = i / wpr;
        const uint32_t w = (uint32_t)(i - row * wpr);
        uint64_t v = 0;
        if (row < pd.sig) {
            const uint64_t fb = a.col0 + (uint64_t)w * 8u;     // file-level byte of this word
            // local bytes beyond valid_bytes and file bytes beyond the row are zero
            const uint64_t gw = fb >> 3;
            uint64_t x = synth_word(a.seed, fpage, row, gw);
            if ((fb & 7u) != 0) {      // column shard not 8-byte aligned: stitch two words
                const uint64_t x2 = synth_word(a.seed, fpage, row, gw + 1);
                const uint32_t s = (uint32_t)(fb & 7u) * 8u;
                x = (x >> s) | (x2 << (64 - s));
            }
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                const uint64_t lb = (uint64_t)w * 8u + b;      // local byte
                const uint64_t gb = fb + b;                    // file-level byte
                uint32_t byte = (uint32_t)(x >> (8 * b)) & 0xFFu;
                if (lb >= pd.valid_bytes || gb >= a.row_bytes || gb * 8 >= live) byte = 0;
                else if (gb * 8 + 8 > live) byte &= (1u << (uint32_t)(live - gb * 8)) - 1u;
                v |= (uint64_t)byte << (8 * b);
            }
        }
        dst[i] = v;
    }
}

// cobs_gpu_plant: one thread per term of the text.  Document i holds term t iff mix64(salt ^ doc << 32 ^ t) % 1000 <
// keep_permille (the test suite's checker restates this rule); a held term sets, for each of its H hashes, bit doc % 8
// of byte doc / 8 of row hash % S_p -- what classic_index.cpp:40-73 does for a document's own terms.
__global__ __launch_bounds__(256) void plant_kernel(PlantArgs a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = a.term_size;
    if (a.len < k || t > a.len - k) return;
    const uint8_t* text = a.text + t;
    KmerView kv{text, k, 0u};
    if (a.canonicalize != 0) {
        for (uint32_t s = 0; s < k; ++s)
            if (fwd_base(text[s]) == 0) { *a.bad = 1u; return; }
        uint32_t mode = 1;
        for (uint32_t s = 0; s < k / 2; ++s) {
            const int f = (int)fwd_base(text[s]);
            const int r = (int)rev_base(text[k - 1 - s]);
            if (f < r) break;
            if (f > r) { mode = 2; break; }
        }
        kv.mode = mode;
    }
    for (uint32_t j = 0; j < a.num_hashes; ++j) {
        const uint64_t h = xxh64_view(kv, (uint64_t)j);
        for (uint32_t i = 0; i < a.ndocs; ++i) {
            const PlantDoc d = a.docs[i];
            if (!d.col) continue;
            if (mix64(a.salt ^ ((uint64_t)d.doc << 32) ^ (uint64_t)t) % 1000u >= d.keep_permille) continue;
            uint8_t* byte = d.col + (h % d.sig) * (uint64_t)d.pitch;
            const uintptr_t addr = reinterpret_cast<uintptr_t>(byte);
            atomicOr(reinterpret_cast<uint32_t*>(addr & ~(uintptr_t)3), 1u << (8u * (uint32_t)(addr & 3u) + d.bit));
        }
    }
}

// rows [row0, row0 + nrows) of one sub-index of the procedural index, packed `pitch` bytes apart
// (the file writer: cobs_gpu_write_synthetic)
__global__ __launch_bounds__(256) void synth_rows_kernel(SynthRowsArgs a) {
    const uint32_t wpr = a.pitch / 8u;
    const uint64_t nwords = a.nrows * (uint64_t)wpr;
    uint64_t* dst = reinterpret_cast<uint64_t*>(a.dst);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / wpr;
        const uint32_t w = (uint32_t)(i - r * wpr);
        const uint64_t x = synth_word(a.seed, a.page, a.row0 + r, w);
        uint64_t v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const uint64_t gb = (uint64_t)w * 8u + b;          // row byte
            uint32_t byte = (uint32_t)(x >> (8 * b)) & 0xFFu;
            if (gb >= a.row_bytes || gb * 8 >= a.live_docs) byte = 0;
            else if (gb * 8 + 8 > a.live_docs) byte &= (1u << (uint32_t)(a.live_docs - gb * 8)) - 1u;
            v |= (uint64_t)byte << (8 * b);
        }
        dst[i] = v;
    }
}

// staged raw rows -> pitched rows (16 bytes per thread), zero padding to the pitch
__global__ __launch_bounds__(256) void repitch_kernel(RepitchArgs a) {
    const uint32_t cpr = a.dst_pitch / 16u;
    const uint64_t total = a.rows * cpr;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = i / cpr;
        const uint32_t c = (uint32_t)(i - row * cpr);
        const uint8_t* s = a.src + row * a.src_pitch + a.src_col0 + (uint64_t)c * 16u;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t b = 0; b < 16; ++b) {
            if (c * 16u + b < a.copy_bytes) w[b >> 2] |= (uint32_t)s[b] << (8 * (b & 3u));
        }
        *reinterpret_cast<uint4*>(a.dst + row * a.dst_pitch + (uint64_t)c * 16u) =
            make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------
// launchers

hipError_t launch_hash(const HashArgs& a, uint64_t total_threads, hipStream_t stream) {
    if (total_threads == 0) return hipSuccess;
    const uint64_t blocks = (total_threads + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (a.term_size == 31) {
        if (a.idx64) hipLaunchKernelGGL(hash_kernel_k31<uint64_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a, total_threads);
        else hipLaunchKernelGGL(hash_kernel_k31<uint32_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a, total_threads);
    } else {
        if (a.idx64) hipLaunchKernelGGL(hash_kernel<uint64_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a, total_threads);
        else hipLaunchKernelGGL(hash_kernel<uint32_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a, total_threads);
    }
    return hipGetLastError();
}

template <int NP, int NW, bool H1, typename OutT, bool MQ = false, typename IdxT = uint32_t, bool LDSS = false, bool TK = false>
static hipError_t launch_scan_inst(const ScanArgs& a, uint32_t ntiles, hipStream_t stream) {
    (void)ntiles;
    const uint32_t per_group = MQ ? 64u / a.tile_w : 1u;
    const uint64_t groups = (uint64_t)((a.chunk_end - a.chunk_begin + a.tile_w - 1) / a.tile_w) *
                            ((a.nq + per_group - 1u) / per_group);
    if (groups == 0) return hipSuccess;
    if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
    constexpr size_t front = scan_lds_front<NP, NW, sizeof(OutT), LDSS>();
    constexpr size_t lds = front + (sizeof(OutT) == 1 ? 0 : 256 * sizeof(uint4)) + 64 * 4 * sizeof(uint32_t);
    auto kern = scan_kernel<NP, NW, H1, OutT, MQ, IdxT, LDSS, TK>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)groups), dim3(NW * 64), lds, stream, a);
    return hipGetLastError();
}

// multi-query variant: H = 1, u16 scores (short queries)
template <int NP, typename OutT>
static hipError_t launch_scan_mq(const ScanArgs& a, uint32_t ntiles, int nw, hipStream_t stream) {
    if (a.cand) {           // run_topk without score rows
        if (nw == 1) return launch_scan_inst<NP, 1, true, OutT, true, uint32_t, false, true>(a, ntiles, stream);
        if (nw == 2) return launch_scan_inst<NP, 2, true, OutT, true, uint32_t, false, true>(a, ntiles, stream);
        return launch_scan_inst<NP, 4, true, OutT, true, uint32_t, false, true>(a, ntiles, stream);
    }
    if (nw == 1) return launch_scan_inst<NP, 1, true, OutT, true>(a, ntiles, stream);
    if (nw == 2) return launch_scan_inst<NP, 2, true, OutT, true>(a, ntiles, stream);
    return launch_scan_inst<NP, 4, true, OutT, true>(a, ntiles, stream);
}

template <int NP, typename OutT>
static hipError_t launch_scan_np(const ScanArgs& a, uint32_t ntiles, bool h1, int nw, hipStream_t stream) {
    if (a.cand) {           // run_topk without score rows: 32-bit row indices (scan_has_tile_topk)
        if (a.idx64) return hipErrorInvalidValue;
        if (nw == 1) return h1 ? launch_scan_inst<NP, 1, true, OutT, false, uint32_t, false, true>(a, ntiles, stream)
                               : launch_scan_inst<NP, 1, false, OutT, false, uint32_t, false, true>(a, ntiles, stream);
        if (nw == 2) return h1 ? launch_scan_inst<NP, 2, true, OutT, false, uint32_t, false, true>(a, ntiles, stream)
                               : launch_scan_inst<NP, 2, false, OutT, false, uint32_t, false, true>(a, ntiles, stream);
        return h1 ? launch_scan_inst<NP, 4, true, OutT, false, uint32_t, false, true>(a, ntiles, stream)
                  : launch_scan_inst<NP, 4, false, OutT, false, uint32_t, false, true>(a, ntiles, stream);
    }
    if (a.idx64) {
        // sub-indexes with >= 2^32 rows: 64-bit row indices; two waves per group cover every
        // query length well enough for this rare geometry (keeps the instantiation count down)
        return h1 ? launch_scan_inst<NP, 2, true, OutT, false, uint64_t>(a, ntiles, stream)
                  : launch_scan_inst<NP, 2, false, OutT, false, uint64_t>(a, ntiles, stream);
    }
    if (nw == 1)
        return h1 ? launch_scan_inst<NP, 1, true, OutT>(a, ntiles, stream)
                  : launch_scan_inst<NP, 1, false, OutT>(a, ntiles, stream);
    if (nw == 2)
        return h1 ? launch_scan_inst<NP, 2, true, OutT>(a, ntiles, stream)
                  : launch_scan_inst<NP, 2, false, OutT>(a, ntiles, stream);
    return h1 ? launch_scan_inst<NP, 4, true, OutT>(a, ntiles, stream)
              : launch_scan_inst<NP, 4, false, OutT>(a, ntiles, stream);
}

int scan_planes_for(uint64_t max_terms) {
    int need = 1;
    while (need < 64 && (max_terms >> need) != 0) ++need;     // bit width of max_terms
    static const int avail[] = {4, 8, 10, 12, 16, 20, 24, 32};
    for (int v : avail)
        if (v >= need) return v;
    return -1;
}

bool scan_has_multi_query(int planes, uint32_t num_hashes, uint32_t tile_w) {
    return num_hashes == 1 && tile_w < 64 && (planes == 4 || planes == 8 || planes == 10 || planes == 12);
}

bool scan_has_tile_topk(uint32_t num_hashes, bool idx64) { (void)num_hashes; return !idx64; }

bool scan_has_lds_staged(int planes, uint32_t num_hashes, int nw) {
    return num_hashes == 1 && planes == 10 && (nw == 2 || nw == 4);
}

hipError_t launch_scan(const ScanArgs& a, uint32_t ntiles, int planes, int nw, bool multi_query,
                       hipStream_t stream) {
    const bool h1 = a.num_hashes == 1;
    if (a.cand && (a.lds_staged || a.topk_k == 0 || !scan_has_tile_topk(a.num_hashes, a.idx64 != 0))) return hipErrorInvalidValue;
    if (a.lds_staged) {     // measured variant (A/B): rows through LDS
        if (multi_query || a.idx64 || !scan_has_lds_staged(planes, a.num_hashes, nw)) return hipErrorInvalidValue;
        return nw == 2 ? launch_scan_inst<10, 2, true, uint16_t, false, uint32_t, true>(a, ntiles, stream)
                       : launch_scan_inst<10, 4, true, uint16_t, false, uint32_t, true>(a, ntiles, stream);
    }
    if (multi_query) {
        if (a.idx64 || !scan_has_multi_query(planes, a.num_hashes, a.tile_w)) return hipErrorInvalidValue;
        switch (planes) {
        case 4: return launch_scan_mq<4, uint8_t>(a, ntiles, nw, stream);
        case 8: return launch_scan_mq<8, uint8_t>(a, ntiles, nw, stream);
        case 10: return launch_scan_mq<10, uint16_t>(a, ntiles, nw, stream);
        default: return launch_scan_mq<12, uint16_t>(a, ntiles, nw, stream);
        }
    }
    switch (planes) {
    case 4: return launch_scan_np<4, uint8_t>(a, ntiles, h1, nw, stream);
    case 8: return launch_scan_np<8, uint8_t>(a, ntiles, h1, nw, stream);
    case 10: return launch_scan_np<10, uint16_t>(a, ntiles, h1, nw, stream);
    case 12: return launch_scan_np<12, uint16_t>(a, ntiles, h1, nw, stream);
    case 16: return launch_scan_np<16, uint16_t>(a, ntiles, h1, nw, stream);
    case 20: return launch_scan_np<20, uint32_t>(a, ntiles, h1, nw, stream);
    case 24: return launch_scan_np<24, uint32_t>(a, ntiles, h1, nw, stream);
    case 32: return launch_scan_np<32, uint32_t>(a, ntiles, h1, nw, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_topk(const TopkArgs& a, hipStream_t stream) {
    if (a.nq == 0 || a.k == 0) return hipSuccess;
    if (a.level_bits == 0 || a.level_bits > 12 || a.levels == 0 || a.levels * a.level_bits < a.score_bits ||
        (a.levels - 1) * a.level_bits >= a.score_bits)
        return hipErrorInvalidValue;
    const uint32_t nb = 1u << a.level_bits;
    size_t lds = (size_t)(4 * nb + 256 + 16) * sizeof(uint32_t);
    // the sort reuses the histogram area: 8 bytes per survivor
    uint32_t m = 2;
    while (m < a.sort_limit) m <<= 1;
    if (a.sort_limit) lds = std::max(lds, (size_t)m * 8);
    auto kern = a.from_pool ? topk_kernel<uint32_t, true>
              : a.score_bytes == 1 ? topk_kernel<uint8_t> : a.score_bytes == 2 ? topk_kernel<uint16_t> : topk_kernel<uint32_t>;
    if (a.score_bytes != 1 && a.score_bytes != 2 && a.score_bytes != 4) return hipErrorInvalidValue;
    if (a.from_pool && ((a.counts_stride & 7u) != 0u || a.counts_stride < a.nslots)) return hipErrorInvalidValue;   // 64-byte rows
    if (lds > 64 * 1024 + 2048) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.nq), dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_select_rows(const SelectRowsArgs& a, hipStream_t stream) {
    if (a.nq == 0 || a.nslots == 0) return hipSuccess;
    if ((a.stride % 8u) != 0 || (a.elem_bytes != 1 && a.elem_bytes != 2 && a.elem_bytes != 4)) return hipErrorInvalidValue;
    const uint64_t items = (uint64_t)a.nq * ((a.nslots + 7u) / 8u);
    const uint64_t blocks = (items + 255u) / 256u;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    auto kern = a.elem_bytes == 1 ? select_rows_kernel<uint8_t> : a.elem_bytes == 2 ? select_rows_kernel<uint16_t> : select_rows_kernel<uint32_t>;
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_build(const BuildArgs& a, uint64_t total_bytes, hipStream_t stream) {
    if (total_bytes == 0) return hipSuccess;
    const uint64_t blocks = (total_bytes + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(build_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a, total_bytes);
    return hipGetLastError();
}

hipError_t launch_pack_bytemap(const PackArgs& a, hipStream_t stream) {
    if (a.ndocs == 0 || a.rows == 0) return hipSuccess;
    const uint32_t nwords = ((a.col_base + a.ndocs - 1u) >> 5) - (a.col_base >> 5) + 1u;
    const uint64_t threads = (a.rows + 3u) / 4u * nwords;
    const uint64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_bytemap_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_random_build(const RandomBuildArgs& a, uint64_t ndocs, hipStream_t stream) {
    const uint64_t total = ndocs * a.document_size;
    if (total == 0) return hipSuccess;
    const uint64_t blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(random_build_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a, total);
    return hipGetLastError();
}

hipError_t launch_combine(const CombineArgs& a, hipStream_t stream) {
    if (a.rows == 0 || a.dst_row_bytes == 0) return hipSuccess;
    hipLaunchKernelGGL(combine_kernel, dim3(8192), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_synth(const SynthArgs& a, hipStream_t stream) {
    if (a.npages == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_kernel, dim3(2048, a.npages), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_plant(const PlantArgs& a, hipStream_t stream) {
    if (a.len < a.term_size || a.ndocs == 0) return hipSuccess;
    const uint32_t terms = a.len - a.term_size + 1;
    hipLaunchKernelGGL(plant_kernel, dim3((terms + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_synth_rows(const SynthRowsArgs& a, hipStream_t stream) {
    if (a.nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_rows_kernel, dim3(4096), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_repitch(const RepitchArgs& a, hipStream_t stream) {
    if (a.rows == 0) return hipSuccess;
    const uint64_t total = a.rows * (a.dst_pitch / 16u);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(repitch_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cobs_amd
