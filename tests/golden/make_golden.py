"""Regenerates the golden fixtures of tests/golden/ (run from the repo root:
`python tests/golden/make_golden.py`).

Inputs : tests/golden/fasta/sample{1..7}.fasta[.gz] -- the reference's own test
         corpus (its tests/data/fasta, data files only).
Outputs: c1.cobs_classic / c1.cobs_compact -- the index files that the reference's
         python/tests/test_cobs_index.py:22-61 builds with default parameters
         (k=31, canonicalize=1, 1 hash, fpr 0.3), produced here by the numpy
         construction restatement oracle/construct.py, and
         expected.json -- the known answers they are checked against.

Cross-checks against facts recorded from a run of the real reference during the
survey: file sizes 8864 / 70112 bytes, signature size
8748, data offsets 116 / 128, compact page_size 8 with one sub-index.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import construct as K  # noqa: E402
from oracle import oracle as O  # noqa: E402

Q50 = b"AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"
SHA256 = {"c1.cobs_classic": "3861d48c2bc4ea301111cd0b221b39f6833b3d05a076988c313b37ca73306266",
          "c1.cobs_compact": "2e3a36b5488de5b219d40ef9a6968d4715d81ed9bb950fa22962913d6377f929"}


def main():
    docs = K.fasta_dir_docs(os.path.join(HERE, "fasta"))
    pc = os.path.join(HERE, "c1.cobs_classic")
    pk = os.path.join(HERE, "c1.cobs_compact")
    sig = K.classic_construct(docs, pc)
    page_size, params = K.compact_construct(docs, pk)
    assert sig == 8748 and os.path.getsize(pc) == 8864
    assert page_size == 8 and params == [(8748, 1)] and os.path.getsize(pk) == 70112
    # any drift of oracle/construct.py is loud: the committed fixtures are these exact bytes
    import hashlib
    for p, want in ((pc, SHA256["c1.cobs_classic"]), (pk, SHA256["c1.cobs_compact"])):
        got = hashlib.sha256(open(p, "rb").read()).hexdigest()
        assert got == want, (p, got)
    exp = {"query": Q50.decode(), "classic": {}, "compact": {}}
    for key, p in (("classic", pc), ("compact", pk)):
        ix = O.Index.open(p)
        exp[key]["data_offset"] = ix.data_offset
        exp[key]["counts"] = [int(x) for x in ix.counts(Q50)]
        exp[key]["ranked"] = [[n, s] for (_, _, n, s) in O.search(ix, Q50)]
        exp[key]["thresholds"] = {str(t): [[n, s] for (_, _, n, s) in O.search(ix, Q50, t)]
                                  for t in (0.05, 0.051, 0.15, 0.1500001, 0.8)}
        exp[key]["single_kmer"] = [[n, s] for (_, _, n, s) in O.search(ix, Q50[5:36])]
    a, b = O.Index.open(pk), O.Index.open(pc)
    exp["two_indexes"] = [[n, s] for (_, _, n, s) in O.search([a, b], Q50)]
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
    print("wrote", pc, pk)


if __name__ == "__main__":
    main()
