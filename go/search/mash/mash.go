// Package mash: drop-in for github.com/bebop/poly/search/mash (search/mash/mash.go:52-140) over libpolyhip.
// Exported identifiers, field names and behaviour are the reference's; SketchBatch and DistanceMatrix are the
// additive batch entry points (a single-sequence GPU call cannot win; SURVEY.md 8b).  UNCOMPILED here.
package mash

import "github.com/bebop/poly/internal/polyhip"

// Mash is mash.go:52-56: every field exported and mutable (tests poke Sketches[0], mash_test.go:27).
type Mash struct {
	KmerSize   int
	SketchSize int
	Sketches   []uint32
}

// New is mash.go:59-65.
func New(kmerSize, sketchSize int) *Mash {
	return &Mash{KmerSize: kmerSize, SketchSize: sketchSize, Sketches: make([]uint32, sketchSize)}
}

// Sketch is mash.go:68-104: updates m.Sketches in place (prior state survives where the reference leaves it).
func (m *Mash) Sketch(sequence string) {
	buf, offs := polyhip.Pack([]string{sequence})
	if err := polyhip.MashSketchBatch(buf, offs, m.KmerSize, m.SketchSize, m.Sketches); err != nil {
		panic(err) // the reference has no error return; a device failure is not recoverable here
	}
}

// Similarity is mash.go:107-135.
func (m *Mash) Similarity(other *Mash) float64 {
	counts := make([]uint16, 1)
	if err := polyhip.MashDistanceMatrix(m.Sketches, 1, m.SketchSize, other.Sketches, 1, other.SketchSize, counts, nil); err != nil {
		panic(err)
	}
	smaller := m.SketchSize
	if other.SketchSize < smaller {
		smaller = other.SketchSize
	}
	return float64(counts[0]) / float64(smaller)
}

// Distance is mash.go:138-140.
func (m *Mash) Distance(other *Mash) float64 { return 1 - m.Similarity(other) }

// SketchBatch sketches many sequences in one device call.
func SketchBatch(seqs []string, kmerSize, sketchSize int) []*Mash {
	buf, offs := polyhip.Pack(seqs)
	out := make([]uint32, len(seqs)*sketchSize)
	if err := polyhip.MashSketchBatch(buf, offs, kmerSize, sketchSize, out); err != nil {
		panic(err)
	}
	res := make([]*Mash, len(seqs))
	for i := range seqs {
		res[i] = &Mash{KmerSize: kmerSize, SketchSize: sketchSize, Sketches: out[i*sketchSize : (i+1)*sketchSize : (i+1)*sketchSize]}
	}
	return res
}

// DistanceMatrix returns dist[i*len(ms)+j] = ms[i].Distance(ms[j]) for sketches of one SketchSize.
func DistanceMatrix(ms []*Mash) []float64 {
	if len(ms) == 0 {
		return nil
	}
	s := ms[0].SketchSize
	flat := make([]uint32, 0, len(ms)*s)
	for _, m := range ms {
		flat = append(flat, m.Sketches...)
	}
	dist := make([]float64, len(ms)*len(ms))
	if err := polyhip.MashDistanceMatrix(flat, len(ms), s, flat, len(ms), s, nil, dist); err != nil {
		panic(err)
	}
	return dist
}
