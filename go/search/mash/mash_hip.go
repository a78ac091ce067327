// Drop-in overlay of github.com/bebop/poly/search/mash (search/mash/mash.go:52-140) over libpolyhip.
//
// The fork keeps the reference's mash.go with two declaration renames (go/fork.sh): (*Mash).Sketch -> sketchCPU and
// (*Mash).Similarity -> similarityCPU.  type Mash, New and Distance stay the reference's code; this file supplies the
// two exported methods again -- small inputs run the reference's body, large ones the device -- and the additive batch
// entry points (a single-sequence GPU call cannot win; SURVEY.md 8b).  UNCOMPILED in the authoring image.
package mash

import (
	"fmt"

	"github.com/bebop/poly/internal/polyhip"
)

// Sketch is mash.go:68-104: updates mash.Sketches in place (prior state survives where the reference leaves it).
func (mash *Mash) Sketch(sequence string) {
	if len(sequence) < polyhip.MinSketchBytes {
		mash.sketchCPU(sequence) // the reference's body (BASELINE configs[0]: phiX174 on the CPU)
		return
	}
	buf, offs := polyhip.Pack([]string{sequence})
	if err := polyhip.MashSketchBatch(buf, offs, mash.KmerSize, mash.SketchSize, mash.Sketches); err != nil {
		panic(err) // the reference has no error return; a device failure is not recoverable here
	}
}

// Similarity is mash.go:107-135.  One pair is at most SketchSize merge steps: the reference's body; many pairs go
// through DistanceMatrix.  (Distance, mash.go:138-140, stays the reference's and calls this.)
func (mash *Mash) Similarity(other *Mash) float64 {
	return mash.similarityCPU(other)
}

// SketchBatch sketches many sequences in one device call.
func SketchBatch(seqs []string, kmerSize, sketchSize int) []*Mash {
	buf, offs := polyhip.Pack(seqs)
	out := make([]uint32, len(seqs)*sketchSize+1)
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
	dist := make([]float64, len(ms)*len(ms))
	if len(ms)*len(ms) < polyhip.MinDistancePairs {
		for i, a := range ms {
			for j, b := range ms {
				dist[i*len(ms)+j] = a.Distance(b)
			}
		}
		return dist
	}
	s := ms[0].SketchSize
	flat := make([]uint32, 0, len(ms)*s)
	for _, m := range ms {
		flat = append(flat, m.Sketches...)
	}
	if err := polyhip.MashDistanceMatrix(flat, len(ms), s, flat, len(ms), s, nil, dist); err != nil {
		panic(err)
	}
	return dist
}

// SketchDistanceMatrix is BASELINE configs[2] in one call: New(kmerSize, sketchSize).Sketch(seq) for every sequence
// (mash.go:59-104) and dist[i*n+j] = ms[i].Distance(ms[j]) (mash.go:107-140) without the sketches leaving HBM in between.
// With polyhip.SetDevices / POLYHIP_DEVICES the reads shard over the node's GPUs, which exchange their sketches over
// xGMI and join one block of rows each.
func SketchDistanceMatrix(seqs []string, kmerSize, sketchSize int) ([]*Mash, []float64) {
	n := len(seqs)
	if n == 0 {
		return nil, nil
	}
	buf, offs := polyhip.Pack(seqs)
	sk := make([]uint32, n*sketchSize+1)
	dist := make([]float64, n*n)
	if err := polyhip.MashSketchDistanceMatrix(buf, offs, kmerSize, sketchSize, sk, nil, dist); err != nil {
		panic(err)
	}
	ms := make([]*Mash, n)
	for i := range ms {
		ms[i] = &Mash{KmerSize: kmerSize, SketchSize: sketchSize, Sketches: sk[i*sketchSize : (i+1)*sketchSize : (i+1)*sketchSize]}
	}
	return ms, dist
}

// SharedCounts is an all-vs-all result kept as the INTEGERS the reference divides (mash.go:134:
// float64(sameHashes) / float64(smallerSketch.SketchSize)): n*n uint16 -- 20 GB at BASELINE configs[2]'s 100,000 sketches,
// where the float64 matrix DistanceMatrix / SketchDistanceMatrix return would be 80 GB.  Nothing is lost: Similarity and
// Distance of a pair are derived on demand, bit for bit what (*Mash).Similarity / Distance return.
type SharedCounts struct {
	N          int      // sketches on either side
	SketchSize int      // the (common) SketchSize: the divisor of mash.go:134
	Counts     []uint16 // Counts[i*N+j] = sameHashes of ms[i].Similarity(ms[j])
}

// Same is sameHashes of the pair (mash.go:108,121-123).
func (c *SharedCounts) Same(i, j int) int { return int(c.Counts[i*c.N+j]) }

// Similarity is ms[i].Similarity(ms[j]) (mash.go:134): one float64 division of two integers, as in the reference.
func (c *SharedCounts) Similarity(i, j int) float64 {
	return float64(c.Counts[i*c.N+j]) / float64(c.SketchSize)
}

// Distance is ms[i].Distance(ms[j]) (mash.go:138-140).
func (c *SharedCounts) Distance(i, j int) float64 { return 1 - c.Similarity(i, j) }

// Row is sketch i's counts against every sketch (a view, not a copy).
func (c *SharedCounts) Row(i int) []uint16 { return c.Counts[i*c.N : (i+1)*c.N : (i+1)*c.N] }

// SharedCountsMatrix is DistanceMatrix without the float64 matrix: the shared-hash counts of every ordered pair of
// sketches of one SketchSize (<= 65535, what a uint16 holds).  Small inputs run the reference's merge.
func SharedCountsMatrix(ms []*Mash) *SharedCounts {
	if len(ms) == 0 {
		return &SharedCounts{}
	}
	n, s := len(ms), ms[0].SketchSize
	// one SketchSize for the whole set, and one a uint16 holds: anything else has no meaning as a count matrix (round-5 advice:
	// the restriction was stated, not enforced)
	if s > 65535 {
		panic(fmt.Sprintf("mash.SharedCountsMatrix: SketchSize %d does not fit a uint16 count", s))
	}
	for i, m := range ms {
		if m.SketchSize != s || len(m.Sketches) != s {
			panic(fmt.Sprintf("mash.SharedCountsMatrix: sketch %d has SketchSize %d, the set's is %d", i, m.SketchSize, s))
		}
	}
	res := &SharedCounts{N: n, SketchSize: s, Counts: make([]uint16, n*n)}
	if n*n < polyhip.MinDistancePairs {
		for i, a := range ms {
			for j, b := range ms {
				res.Counts[i*n+j] = uint16(sharedHashes(a.Sketches, b.Sketches)) // an integer merge, no float64 round trip
			}
		}
		return res
	}
	flat := make([]uint32, 0, n*s)
	for _, m := range ms {
		flat = append(flat, m.Sketches...)
	}
	if err := polyhip.MashDistanceMatrix(flat, n, s, flat, n, s, res.Counts, nil); err != nil {
		panic(err)
	}
	return res
}

// sharedHashes counts the hashes two ascending sketches share: the merge of (*Mash).Similarity (mash.go:107-135) without its
// final division.
func sharedHashes(a, b []uint32) int {
	same, i, j := 0, 0, 0
	for i < len(a) && j < len(b) {
		switch {
		case a[i] == b[j]:
			same++
			i++
			j++
		case a[i] < b[j]:
			i++
		default:
			j++
		}
	}
	return same
}

// SketchSharedCounts is BASELINE configs[2] at size in one call: every sequence sketched (mash.go:59-104) and the shared
// counts of every ordered pair (mash.go:107-135), the sketches staying in HBM in between -- SketchDistanceMatrix with the
// n*n uint16 counts in place of n*n float64 (at 100,000 sketches: 20 GB, not 80).  With polyhip.SetDevices /
// POLYHIP_DEVICES the reads shard over the node's GPUs (see SketchDistanceMatrix).
func SketchSharedCounts(seqs []string, kmerSize, sketchSize int) ([]*Mash, *SharedCounts) {
	n := len(seqs)
	if n == 0 {
		return nil, &SharedCounts{}
	}
	buf, offs := polyhip.Pack(seqs)
	sk := make([]uint32, n*sketchSize+1)
	res := &SharedCounts{N: n, SketchSize: sketchSize, Counts: make([]uint16, n*n)}
	if err := polyhip.MashSketchDistanceMatrix(buf, offs, kmerSize, sketchSize, sk, res.Counts, nil); err != nil {
		panic(err)
	}
	ms := make([]*Mash, n)
	for i := range ms {
		ms[i] = &Mash{KmerSize: kmerSize, SketchSize: sketchSize, Sketches: sk[i*sketchSize : (i+1)*sketchSize : (i+1)*sketchSize]}
	}
	return ms, res
}
