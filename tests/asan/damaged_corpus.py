import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import documents as D
from tests.test_documents import _write_corpus
out = os.path.join(sys.argv[2], 'corpus')
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(int(sys.argv[1]))
docdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'golden', 'documents')
srcs = []
for sub in ("cortex", "fastq", "fasta_multi", "text", "../fasta"):
    d = os.path.normpath(os.path.join(docdir, sub))
    for fn in sorted(os.listdir(d)):
        if fn == "document_sorted.txt": continue
        srcs.append(os.path.join(d, fn))
gen = os.path.join(sys.argv[2], 'gen')
_write_corpus(rng, gen)
srcs += [os.path.join(gen, f) for f in sorted(os.listdir(gen))]
n = 0
for src in srcs:
    raw = open(src, 'rb').read()
    base = os.path.basename(src)
    ext = base[base.index('.'):]
    for trial in range(30):
        b = bytearray(raw)
        mode = trial % 5
        if mode == 0: b = b[:int(rng.integers(0, len(b) + 1))]
        elif mode == 1 and len(b):
            for _ in range(int(rng.integers(1, 9))): b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif mode == 2 and len(b):
            at = int(rng.integers(0, min(len(b), 64)))
            b[at:at + 4] = (0xFFFFFFFF if trial % 10 == 2 else int(rng.integers(0, 2 ** 31))).to_bytes(4, 'little')
        elif mode == 3:
            b = b[:int(rng.integers(0, 80))] + bytes(rng.integers(0, 256, size=int(rng.integers(0, 50))).astype(np.uint8))
        else:
            pass   # the intact file
        with open(os.path.join(out, 'd%05d%s' % (n, ext)), 'wb') as f: f.write(bytes(b))
        n += 1
print(n)
