// Package seqhash: RotateSequence of github.com/bebop/poly/seqhash (seqhash.go:78-138) over libpolyhip.
// Hash (seqhash.go:141-224) keeps the reference's host code (upper-casing, validation, BLAKE3 via
// lukechampine.com/blake3) and only its two RotateSequence calls (:182,:188) change callee.  UNCOMPILED here.
package seqhash

import "github.com/bebop/poly/internal/polyhip"

// RotateSequence is seqhash.go:127-138.
func RotateSequence(sequence string) string {
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
