// Package fasta: additive entry point for github.com/bebop/poly/io/fasta.  The reference's Parser
// (fasta.go:90-238) stays as it is; PackAll parses a whole image ON THE DEVICE into the packed batch the GPU
// kernels take, with (*Parser).ParseAll's record semantics (multi-line records, skipped ';' and empty lines,
// the '>'-directly-after-a-header quirk, a last unterminated record dropped).  UNCOMPILED here.
package fasta

import (
	"bytes"
	"errors"

	"github.com/bebop/poly/internal/polyhip"
)

// Batch is every record of a FASTA image.
type Batch struct {
	polyhip.Packed
	file []byte
}

func (b Batch) Len() int { return len(b.Offsets) - 1 }

// Sequence is Fasta.Sequence of record i: its lines joined without newlines (a sub-slice, not a copy).
func (b Batch) Sequence(i int) []byte { return b.Seqs[b.Offsets[i]:b.Offsets[i+1]] }

// Name is Fasta.Name of record i: the header line without '>' (fasta.go:211).
func (b Batch) Name(i int) string {
	line := b.file[b.RecStart[i]:]
	if nl := bytes.IndexByte(line, '\n'); nl >= 0 {
		line = line[:nl]
	}
	return string(line[1:])
}

// PackAll parses the whole image; errors are the reference's (fasta.go:223,227), with the records before
// the bad one in the batch.
func PackAll(file []byte) (Batch, error) {
	p, err := polyhip.FastaPack(file)
	if err != nil {
		return Batch{}, err
	}
	b := Batch{Packed: p, file: file}
	switch p.Code {
	case 0:
		return b, nil
	case 1:
		return b, errors.New("did not find fasta start '>'")
	case 2:
		return b, errors.New("empty fasta sequence")
	}
	return b, errors.New("fasta: device parser error")
}
