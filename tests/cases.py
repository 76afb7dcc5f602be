"""Shared builders of small test indexes (test infrastructure).

Index files are written with oracle/construct.py's header writers, i.e. in the
byte layout of reference cobs/file/{classic,compact}_index_header.cpp, so the same
file is read by the oracle (CPU restatement) and by libcobs_gpu.so.
"""
import os

import numpy as np

from oracle import construct as K
from oracle import oracle as O


def random_bits(rng, shape, density):
    """uint8 matrix with i.i.d. bits of the given density"""
    bits = rng.random((shape[0], shape[1], 8)) < density
    return np.packbits(bits, axis=2, bitorder="little").reshape(shape)


def mask_padding_docs(m, first_doc, num_docs):
    """zero the bits of documents >= num_docs (padding documents carry no bits)"""
    rows, nbytes = m.shape
    live = max(0, min(num_docs - first_doc, nbytes * 8))
    full, rem = divmod(live, 8)
    if full < nbytes:
        if rem:
            m[:, full] &= np.uint8((1 << rem) - 1)
            m[:, full + 1:] = 0
        else:
            m[:, full:] = 0
    return m


def plant(mats, sigs, page_docs, query, docs_fraction, term_size, canonicalize, num_hashes):
    """make every k-mer of `query` present in document d with probability
    docs_fraction[d] (dict doc -> fraction), so thresholds and ranking are exercised"""
    hashes, good = O.term_hashes(query, term_size, canonicalize, num_hashes)
    rng = np.random.default_rng(12345)
    for d, frac in docs_fraction.items():
        p, dd = divmod(d, page_docs) if page_docs else (0, d)
        keep = rng.random(len(hashes)) < frac
        rows = (hashes[keep].reshape(-1) % np.uint64(sigs[p])).astype(np.int64)
        mats[p][rows, dd // 8] |= np.uint8(1 << (dd % 8))


def make_classic(path, num_docs, sig, num_hashes=1, term_size=31, canonicalize=1, density=0.3,
                 seed=1, planted=None, query=None):
    rng = np.random.default_rng(seed)
    row = (num_docs + 7) // 8
    m = mask_padding_docs(random_bits(rng, (sig, row), density), 0, num_docs)
    if planted:
        plant([m], [sig], 0, query, planted, term_size, canonicalize, num_hashes)
    names = ["doc_%05d" % i for i in range(num_docs)]
    K.write_classic(path, term_size, canonicalize, names, sig, num_hashes, m)
    return path


def make_compact(path, num_docs, page_size, sigs, num_hashes=1, term_size=31, canonicalize=1,
                 density=0.3, seed=1, planted=None, query=None):
    rng = np.random.default_rng(seed)
    page_docs = 8 * page_size
    assert (len(sigs) - 1) * page_docs < num_docs <= len(sigs) * page_docs
    mats = []
    for p, s in enumerate(sigs):
        m = random_bits(rng, (s, page_size), density)
        mats.append(mask_padding_docs(m, p * page_docs, num_docs))
    if planted:
        plant(mats, sigs, page_docs, query, planted, term_size, canonicalize, num_hashes)
    names = ["doc_%05d" % i for i in range(num_docs)]
    K.write_compact(path, term_size, canonicalize, page_size, [(s, num_hashes) for s in sigs], names, mats)
    return path


def queries_acgt(n, length, seed):
    return [O.random_sequence(length, seed + i) for i in range(n)]


def oracle_results(ix_list, query, threshold, num_results):
    return [(f, d, s) for (f, d, _name, s) in O.search(ix_list, query, threshold, num_results)]


def tmp(tmp_path, name):
    return os.path.join(str(tmp_path), name)


SURVEY_PROBE_SCORES = [270, 135, 90, 68, 54, 45, 39, 34, 30, 27, 25, 23, 21, 20, 18, 17, 16, 15, 15, 14]


def survey_probe_files(oracle, construct, tmp_path):
    """SURVEY 8c [probed]: the real reference, run during the survey on a hand-built 20-document
    classic (S = 5003, H = 3) and compact (page_size 1, P = 3, S_p = {3001, 4001, 5003}) index,
    returned SURVEY_PROBE_SCORES for random_sequence(300, 7) -- document j holds every (j+1)-th
    term of the query.  -> (classic path, compact path, query, names, expected scores)"""
    q = oracle.random_sequence(300, 7)
    H, k = 3, 31
    hashes, good = oracle.term_hashes(q, k, 1, H)
    assert good.all() and len(hashes) == 270
    names = ["doc_%02d" % j for j in range(20)]
    docs = [construct.Doc(names[j], names[j], 0, len(range(0, 270, j + 1)), hashes[np.arange(0, 270, j + 1)])
            for j in range(20)]
    pc = tmp(tmp_path, "probe.cobs_classic")
    construct.write_classic(pc, k, 1, names, 5003, H, construct.build_matrix(docs, 5003, 3))
    pk = tmp(tmp_path, "probe.cobs_compact")
    sigs = [3001, 4001, 5003]
    mats = [construct.build_matrix(docs[8 * p:8 * p + 8], sigs[p], 1) for p in range(3)]
    construct.write_compact(pk, k, 1, 1, [(s, H) for s in sigs], names, mats)
    return pc, pk, q, names, SURVEY_PROBE_SCORES
