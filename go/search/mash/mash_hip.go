// Drop-in overlay of github.com/bebop/poly/search/mash (search/mash/mash.go:52-140) over libpolyhip.
//
// The fork keeps the reference's mash.go with two declaration renames (go/fork.sh): (*Mash).Sketch -> sketchCPU and
// (*Mash).Similarity -> similarityCPU.  type Mash, New and Distance stay the reference's code; this file supplies the
// two exported methods again -- small inputs run the reference's body, large ones the device -- and the additive batch
// entry points (a single-sequence GPU call cannot win; SURVEY.md 8b).  UNCOMPILED in the authoring image.
package mash

import "github.com/bebop/poly/internal/polyhip"

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
