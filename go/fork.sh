#!/bin/sh
# go/fork.sh POLY_CHECKOUT -- turn a checkout of github.com/bebop/poly into the MI355X drop-in fork.
#
# The reference's files stay where they are; the hot-path entry points are RENAMED in place (declaration lines only) so
# that their bodies remain the small-input path, and the *_hip.go overlays of this directory supply the exported names
# again: below polyhip's size thresholds they call the renamed reference body, above them libpolyhip.so through cgo.
# Nothing else of the reference changes, and no reference source is stored in this repository.
set -eu
P=${1:?usage: go/fork.sh POLY_CHECKOUT}
HERE=$(cd "$(dirname "$0")" && pwd)
ren() { # file, old declaration prefix, new declaration prefix
	grep -q "^$2" "$P/$1" || { echo "fork.sh: '$2' not found in $1 (reference changed?)" >&2; exit 1; }
	sed -i "s|^$2|$3|" "$P/$1"
}
ren search/mash/mash.go   'func (mash \*Mash) Sketch('      'func (mash *Mash) sketchCPU('
ren search/mash/mash.go   'func (mash \*Mash) Similarity('  'func (mash *Mash) similarityCPU('
ren search/align/align.go 'func SmithWaterman('             'func smithWatermanCPU('
ren search/align/align.go 'func NeedlemanWunsch('           'func needlemanWunschCPU('
ren primers/primers.go    'func SantaLucia('                'func santaLuciaCPU('
ren primers/primers.go    'func MarmurDoty('                'func marmurDotyCPU('
ren primers/pcr/pcr.go    'func DesignPrimersWithOverhangs(' 'func designPrimersWithOverhangsCPU('
ren seqhash/seqhash.go    'func Hash('                      'func hashCPU('
ren seqhash/seqhash.go    'func RotateSequence('            'func rotateSequenceCPU('
ren clone/clone.go        'func CircularLigate('            'func circularLigateCPU('
mkdir -p "$P/internal/polyhip"
cp "$HERE"/polyhip/*.go "$P/internal/polyhip/"
for d in search/mash search/align primers primers/pcr seqhash clone io/fasta io/fastq; do
	cp "$HERE"/$d/*_hip.go "$P/$d/"
done
echo "fork.sh: done.  Build with:"
echo "  CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS='-L<repo>/poly_amd -lpolyhip -Wl,-rpath,<repo>/poly_amd' go build ./..."
echo "then follow go/VALIDATE.md"
