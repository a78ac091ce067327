#!/usr/bin/env python3
"""Regenerates the fixture files in tests/golden/ from the reference checkout.

Run in the authoring container only (needs /root/reference); the outputs are
committed so that nothing at test/bench time reads /root/reference.

  phix174.seq  <- data/phix174.gb   (BASELINE config 1 input; 5,386 bp, lower case)
  puc19.seq    <- data/puc19.gbk    (seqhash_test.go:68-91 rotation fixture)
  fastq/*.fastq <- io/fastq/data/*.fastq  (verbatim copies of the reference's own parser fixtures:
                   fastq_test.go:59-66, example_test.go:16-66)
  fasta/base.fasta <- io/fasta/data/base.fasta  (example_test.go:18-36,100-114)
  clone_goldengate_rotated.seq <- the `// Output:` line of clone/example_test.go:11-31 (ExampleGoldenGate prints
                   seqhash.RotateSequence(Clones[0])): a 4.4 kb circular construct already at its least rotation

Extraction follows io/genbank/genbank.go:125,627-633: every line between
ORIGIN and // with all non-letters removed, case preserved.
"""
import os
import re
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def genbank_sequence(path: str) -> str:
    seq, on = [], False
    with open(path) as f:
        for line in f:
            if line.startswith("ORIGIN"):
                on = True
                continue
            if on:
                if line.startswith("//"):
                    break
                seq.append(re.sub(r"[^a-zA-Z]+", "", line))
    return "".join(seq)


def main() -> int:
    for src, dst in (("data/phix174.gb", "phix174.seq"), ("data/puc19.gbk", "puc19.seq")):
        s = genbank_sequence(os.path.join(REF, src))
        with open(os.path.join(HERE, dst), "w") as f:
            f.write(s + "\n")
        print(dst, len(s))
    return 0


def copy_fastq():
    import glob
    import shutil
    dst = os.path.join(HERE, "fastq")
    os.makedirs(dst, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(REF, "io", "fastq", "data", "*.fastq"))):
        shutil.copy(f, dst)
    os.makedirs(os.path.join(HERE, "fasta"), exist_ok=True)
    shutil.copy(os.path.join(REF, "io", "fasta", "data", "base.fasta"), os.path.join(HERE, "fasta"))


def clone_example_output():
    """the expected output of ExampleGoldenGate (clone/example_test.go:30-31)"""
    with open(os.path.join(REF, "clone", "example_test.go")) as f:
        src = f.read()
    at = src.index("fmt.Println(seqhash.RotateSequence(Clones[0]))")
    out = re.search(r"// Output: ([A-Za-z]+)", src[at:]).group(1)
    with open(os.path.join(HERE, "clone_goldengate_rotated.seq"), "w") as f:
        f.write(out + "\n")
    print("clone_goldengate_rotated.seq", len(out))


if __name__ == "__main__":
    copy_fastq()
    clone_example_output()
    sys.exit(main())
