"""Measures GPU index construction end to end and per stage.

    python scripts/construct_bench.py [--docs 256] [--doc-mb 4] [--dir /tmp/cobs_docs] [--cpu-seconds 10]

Writes `--docs` FASTA documents of random bases (`--doc-mb` MB each, 80-column lines) to a scratch
directory, then times, on one GPU:
  list      cobs_gpu_doclist_add_recursive (the index pass over every file: sizes, term counts)
  resident  cobs_gpu_build_index_list -> a query handle (files read + parsed by host threads,
            H2D, build_kernel; no file written)
  file      cobs_gpu_build_classic_list -> .cobs_classic on local disk (adds the D2H stream + write)
  memory    the same documents handed over as in-memory texts (no file parsing: upload + kernel only)
and the checker's hashing of a bounded sample on one host core (the port of process_term,
classic_index.cpp:46-73) as the CPU figure beside it.  One JSON line; `rocprofv3 --kernel-trace
--stats` around this command gives build_kernel's own time (profiles/r02_construct_*)."""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_docs(d, ndocs, doc_bytes, seed=1):
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(ndocs):
        n = int(doc_bytes * (0.5 + rng.random()))            # sizes spread around the mean
        lines = (n + 79) // 80
        grid = np.full((lines, 81), 10, dtype=np.uint8)      # 80 bases + '\n' per line
        flat = np.zeros(lines * 80, dtype=np.uint8)
        flat[:n] = lut[rng.integers(0, 4, size=n, dtype=np.uint8)]
        grid[:, :80] = flat.reshape(lines, 80)
        tail = n - (lines - 1) * 80                           # characters of the last line
        with open(os.path.join(d, "genome_%05d.fasta" % i), "wb") as f:
            f.write(b">genome_%05d random bases\n" % i)
            f.write(grid.reshape(-1)[:(lines - 1) * 81 + tail].tobytes() + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=256)
    ap.add_argument("--doc-mb", type=float, default=4.0)
    ap.add_argument("--dir", default="/tmp/cobs_construct_bench")
    ap.add_argument("--num-hashes", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--batch-mb", type=int, default=0, help="text batch size (0 = the library's default)")
    ap.add_argument("--set-bits-mode", type=int, default=0, help="0 auto, 1 atomics, 2 byte planes")
    a = ap.parse_args()
    import torch  # noqa: F401
    import cobs_amd
    from oracle import oracle as O

    docdir = os.path.join(a.dir, "docs")
    t0 = time.time()
    write_docs(docdir, a.docs, int(a.doc_mb * 1e6))
    t_gen = time.time() - t0
    file_bytes = sum(os.path.getsize(os.path.join(docdir, f)) for f in os.listdir(docdir))

    p = cobs_amd.ClassicIndexParameters()
    p.num_hashes, p.false_positive_rate, p.clobber = a.num_hashes, 0.3, True
    p.text_batch_bytes, p.set_bits_mode = a.batch_mb << 20, a.set_bits_mode
    out = {"docs": a.docs, "file_bytes": file_bytes, "num_hashes": a.num_hashes, "generate_s": round(t_gen, 2)}

    t0 = time.time()
    dl = cobs_amd.DocumentList(docdir)
    out["list_s"] = round(time.time() - t0, 3)
    terms = sum(d.num_terms(31) for d in dl)
    out["terms"] = terms

    cobs_amd.build_search(list=dl, index_params=p).close()              # warm-up: HIP context, pinned pools
    best = {}
    for rep in range(2):
        t0 = time.time()
        s = cobs_amd.build_search(list=dl, index_params=p)
        dt = time.time() - t0
        best["resident_s"] = min(best.get("resident_s", 1e9), dt)
        sig = s.signature_size(0, 0)
        s.close()
        idx = os.path.join(a.dir, "bench.cobs_classic")
        t0 = time.time()
        cobs_amd.classic_construct(list=dl, out_file=idx, index_params=p)
        best["file_s"] = min(best.get("file_s", 1e9), time.time() - t0)
    out["signature_size"] = sig
    out["index_bytes"] = os.path.getsize(idx)

    # the same documents as in-memory texts: what is left when nothing is read or parsed
    mem = cobs_amd.DocumentList()
    for d in dl:
        raw = open(d.path, "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
        mem.add_document(d.name, [raw])
    for rep in range(2):
        t0 = time.time()
        s = cobs_amd.build_search(list=mem, index_params=p)
        best["memory_s"] = min(best.get("memory_s", 1e9), time.time() - t0)
        s.close()
    out.update({k: round(v, 3) for k, v in best.items()})
    for k in ("resident_s", "file_s", "memory_s"):
        out[k.replace("_s", "_Mterms_per_s")] = round(terms / best[k] / 1e6, 1)
        out[k.replace("_s", "_GB_per_s")] = round(file_bytes / best[k] / 1e9, 2)

    # CPU figure: the checker's term hashing (canonicalise + XXH64 per hash) on one core
    seq = O.random_sequence(2_000_000, 3)
    n, t0 = 0, time.time()
    while time.time() - t0 < a.cpu_seconds:
        O.term_hashes(seq, 31, 1, a.num_hashes)
        n += len(seq) - 30
    out["cpu_port_Mterms_per_s_1core"] = round(n / (time.time() - t0) / 1e6, 2)
    out["cpu_note"] = "hashing only (no bit setting, no parsing): an upper bound of the reference's per-core rate"
    print(json.dumps(out))
    if not a.keep:
        shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
