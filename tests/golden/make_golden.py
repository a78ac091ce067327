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

  clone_parts.json <- the clone.Part literals of clone/clone_test.go and clone/example_test.go

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


def clone_parts():
    """the clone.Part literals of the reference's own clone tests (clone/clone_test.go, clone/example_test.go):
    the inputs of TestSignalKilledGoldenGate (:167-194: 1 clone + 4 looping constructs), TestPanicGoldenGate
    (:196-214), TestCircularCutRegression (:216-228) and ExampleGoldenGate (example_test.go:11-31)"""
    import json

    def parts_of(path, start, end):
        with open(os.path.join(REF, "clone", path)) as f:
            lines = f.read().split("\n")[start - 1:end]
        return [[m.group(1), m.group(2) == "true"] for ln in lines for m in re.finditer(r'Part\{"([A-Za-z]+)", (true|false)\}', ln)]

    out = {
        "popen": parts_of("clone_test.go", 8, 8)[0],
        "signal_killed": parts_of("clone_test.go", 167, 179),
        "panic": parts_of("clone_test.go", 196, 206),
        "circular_cut_regression": parts_of("clone_test.go", 216, 222),
        "example_golden_gate": parts_of("example_test.go", 11, 24),
    }
    assert len(out["signal_killed"]) == 9 and len(out["panic"]) == 5 and len(out["example_golden_gate"]) == 3
    with open(os.path.join(HERE, "clone_parts.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("clone_parts.json", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    copy_fastq()
    clone_example_output()
    clone_parts()
    sys.exit(main())
