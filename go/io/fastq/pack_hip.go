// Package fastq: additive entry point for github.com/bebop/poly/io/fastq.  The reference's Parser
// (fastq.go:60-216) stays as it is; PackAll is the bulk path in front of the GPU kernels: the file image is
// parsed ON THE DEVICE into the packed batch mash.SketchPacked / align.SmithWatermanPacked take, with the
// same record semantics and the same errors as (*Parser).ParseAll.  UNCOMPILED here.
package fastq

import (
	"bytes"
	"fmt"
	"strings"

	"github.com/bebop/poly/internal/polyhip"
)

// Batch is every record of a FASTQ image: Sequence(i) without per-record allocations, identifiers on demand.
type Batch struct {
	polyhip.Packed
	file []byte
}

// Len is the number of records parsed (those before the first bad record, like ParseN).
func (b Batch) Len() int { return len(b.Offsets) - 1 }

// Sequence is Fastq.Sequence of record i (a sub-slice of the packed buffer, not a copy).
func (b Batch) Sequence(i int) []byte { return b.Seqs[b.Offsets[i]:b.Offsets[i+1]] }

// Identifier is Fastq.Identifier of record i: the identifier line up to the first space, without '@'
// (fastq.go:158-159).
func (b Batch) Identifier(i int) string {
	line := b.file[b.RecStart[i]:]
	if nl := bytes.IndexByte(line, '\n'); nl >= 0 {
		line = line[:nl]
	}
	id := string(line)
	if sp := strings.IndexByte(id, ' '); sp >= 0 {
		id = id[:sp]
	}
	return id[1:]
}

// PackAll parses the whole image.  The error, if any, carries the reference's message for the first bad
// record (fastq.go:147,177,198,204); the records before it are in the batch.  Two inputs on which the
// reference PANICS (an empty identifier line, fastq.go:156; an identifier field without '=', :163) are
// reported as errors instead.
func PackAll(file []byte) (Batch, error) {
	p, err := polyhip.FastqPack(file)
	if err != nil {
		return Batch{}, err
	}
	b := Batch{Packed: p, file: file}
	switch p.Code {
	case 0:
		return b, nil
	case 1:
		return b, fmt.Errorf("did not find fastq start '@', got to line %d", p.Line)
	case 2:
		return b, fmt.Errorf("empty fastq sequence, got to line %d", p.Line)
	case 3:
		return b, fmt.Errorf("empty quality sequence, got to line %d", p.Line)
	case 4:
		return b, fmt.Errorf("line %d failed: unexepcted EOF encountered", p.Line)
	case 5:
		return b, fmt.Errorf("empty identifier on line %d (the reference parser panics here)", p.Line)
	case 6:
		return b, fmt.Errorf("identifier field without '=' on line %d (the reference parser panics here)", p.Line)
	}
	return b, fmt.Errorf("fastq: device parser code %d", p.Code)
}
