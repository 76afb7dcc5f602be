"""Copies the DATA files of the reference's document-reader tests into tests/golden/documents/
(run in the build container, where /root/reference exists: `python tests/golden/copy_reference_documents.py`).

These are the fixtures of the reference's tests/cortex_file.cpp, fastq_file.cpp,
fasta_multifile.cpp and text_file.cpp (its tests/data/{cortex,fastq,fasta_multi,text}): input
documents plus, for McCortex, the expected k-mer lists (sample1-k*.txt, document_sorted.txt).
Data only -- no source file of the reference is copied."""
import os
import shutil

SRC = "/root/reference/tests/data"
HERE = os.path.dirname(os.path.abspath(__file__))

for sub in ("cortex", "fastq", "fasta_multi", "text"):
    dst = os.path.join(HERE, "documents", sub)
    os.makedirs(dst, exist_ok=True)
    for fn in sorted(os.listdir(os.path.join(SRC, sub))):
        shutil.copyfile(os.path.join(SRC, sub, fn), os.path.join(dst, fn))
        os.chmod(os.path.join(dst, fn), 0o644)
        print(sub, fn)
