// Drop-in overlay of github.com/bebop/poly/seqhash over libpolyhip: RotateSequence (seqhash.go:78-138), Hash (:141-224).
//
// The fork keeps the reference's seqhash.go with two declaration renames (go/fork.sh): Hash -> hashCPU, RotateSequence ->
// rotateSequenceCPU; SequenceType and its constants, boothLeastRotation and the v2 hashes stay the reference's code.
// Small sequences run the reference's body, large ones and the batch entry points the device.  UNCOMPILED here.
package seqhash

import (
	"errors"

	"github.com/bebop/poly/internal/polyhip"
)

// Hash is seqhash.go:141-224 (same seqhash strings, same error texts).
func Hash(sequence string, sequenceType SequenceType, circular bool, doubleStranded bool) (string, error) {
	if len(sequence) < polyhip.MinRotateBytes {
		return hashCPU(sequence, sequenceType, circular, doubleStranded) // the reference's body
	}
	res, errs := HashBatch([]string{sequence}, sequenceType, circular, doubleStranded)
	return res[0], errs[0]
}

// HashBatch hashes many sequences under one (type, circular, doubleStranded) in one device call
// (the shape of clone's dedup loop, clone.go:269-320).
func HashBatch(sequences []string, sequenceType SequenceType, circular bool, doubleStranded bool) ([]string, []error) {
	code := map[SequenceType]int{DNA: 0, RNA: 1, PROTEIN: 2}
	out, errs := make([]string, len(sequences)), make([]error, len(sequences))
	c, ok := code[sequenceType]
	if !ok { // seqhash.go:152
		for i := range errs {
			errs[i] = errors.New("Only sequenceTypes of DNA, RNA, or PROTEIN allowed. Got sequenceType: " + string(sequenceType))
		}
		return out, errs
	}
	ds := doubleStranded && sequenceType != PROTEIN
	// a byte >= 0x80 is not an error of the BATCH: the reference returns an ordinary per-sequence error for it
	// ("Only letters ... Got letter: ...", seqhash.go:157,169, after upper-casing it as UTF-8), and clone.CircularLigate
	// drops that error on purpose.  Those sequences run the reference's own body; the device gets a stand-in.
	packed, odd := sequences, []int(nil)
	for i, q := range sequences {
		if !polyhip.IsASCII(q) {
			if odd == nil {
				packed = append([]string(nil), sequences...)
			}
			packed[i] = "A"
			odd = append(odd, i)
		}
	}
	buf, offs := polyhip.Pack(packed)
	hashes, codes, err := polyhip.SeqhashBatch(buf, offs, c, circular, ds)
	if err != nil {
		panic(err) // a device failure: the reference's signature has nowhere to put it
	}
	isOdd := make(map[int]bool, len(odd))
	for _, i := range odd {
		isOdd[i] = true
		out[i], errs[i] = hashCPU(sequences[i], sequenceType, circular, doubleStranded)
	}
	for i := range sequences {
		switch {
		case isOdd[i]:
		case codes[i]>>8 == 2: // seqhash.go:157
			errs[i] = errors.New("Only letters ATUGCYRSWKMBDHVNZ are allowed for DNA/RNA. Got letter: " + string(rune(codes[i]&0xFF)))
		case codes[i]>>8 == 3: // seqhash.go:169
			errs[i] = errors.New("Only letters ACDEFGHIKLMNPQRSTVWYUO*BXZ are allowed for Proteins. Got letter: " + string(rune(codes[i]&0xFF)))
		case sequenceType == PROTEIN && doubleStranded: // seqhash.go:175
			errs[i] = errors.New("Proteins cannot be double stranded")
		default:
			out[i] = hashes[i]
		}
	}
	return out, errs
}

// RotateSequence is seqhash.go:127-138.
func RotateSequence(sequence string) string {
	if len(sequence) < polyhip.MinRotateBytes {
		return rotateSequenceCPU(sequence) // the reference's body
	}
	return RotateBatch([]string{sequence})[0]
}

// RotateBatch rotates many circular sequences to their least rotation in one device call.
func RotateBatch(seqs []string) []string {
	buf, offs := polyhip.Pack(seqs)
	_, rotated, err := polyhip.LeastRotationBatch(buf, offs)
	if err != nil {
		panic(err)
	}
	out := make([]string, len(seqs))
	for i := range seqs {
		out[i] = string(rotated[offs[i]:offs[i+1]])
	}
	return out
}
