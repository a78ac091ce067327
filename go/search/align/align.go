// Package align: drop-in for github.com/bebop/poly/search/align (align.go:73-95,171-232) over libpolyhip.
// NeedlemanWunsch (align.go:100-166) is not on the north-star path and keeps the reference's Go code.
// UNCOMPILED here.
package align

import (
	"github.com/bebop/poly/alphabet"
	"github.com/bebop/poly/internal/polyhip"
	"github.com/bebop/poly/search/align/matrix"
)

// Scoring is align.go:73-76.
type Scoring struct {
	SubstitutionMatrix *matrix.SubstitutionMatrix
	GapPenalty         int
	dev                *polyhip.Scoring // flattened tables on the device, built on first use
}

// NewScoring is align.go:79-87 (nil matrix -> matrix.Default; never errors).
func NewScoring(substitutionMatrix *matrix.SubstitutionMatrix, gapPenalty int) (Scoring, error) {
	if substitutionMatrix == nil {
		substitutionMatrix = matrix.Default
	}
	return Scoring{SubstitutionMatrix: substitutionMatrix, GapPenalty: gapPenalty}, nil
}

// Score is align.go:89-95.
func (s Scoring) Score(a, b byte) (int, error) {
	return s.SubstitutionMatrix.Score(string(a), string(b))
}

// flatten goes through the matrix's PUBLIC Score(): its score table is unexported (matrix.go:13-17).
func (s *Scoring) handle() *polyhip.Scoring {
	if s.dev != nil {
		return s.dev
	}
	var lut [65536]int32
	var va, vb [256]uint8
	for a := 0; a < 128; a++ {
		if _, err := s.SubstitutionMatrix.FirstAlphabet.Encode(string(rune(a))); err == nil {
			va[a] = 1
		}
		if _, err := s.SubstitutionMatrix.SecondAlphabet.Encode(string(rune(a))); err == nil {
			vb[a] = 1
		}
	}
	for a := 0; a < 128; a++ {
		for b := 0; b < 128; b++ {
			if va[a] == 1 && vb[b] == 1 {
				v, _ := s.SubstitutionMatrix.Score(string(rune(a)), string(rune(b)))
				lut[a*256+b] = int32(v)
			}
		}
	}
	h, err := polyhip.NewScoring(&lut, &va, &vb, s.GapPenalty)
	if err != nil {
		panic(err)
	}
	s.dev = h
	return h
}

func symbolError(code uint32) error {
	return &alphabet.Error{Message: "Symbol " + string(rune(code&0xFF)) + " not in alphabet"} // alphabet.go:38
}

// SmithWaterman is align.go:171-232.
func SmithWaterman(stringA string, stringB string, scoring Scoring) (int, string, string, error) {
	res := SmithWatermanBatch([]string{stringA}, stringB, &scoring)
	if res[0].Err != nil {
		return 0, "", "", res[0].Err
	}
	return res[0].Score, res[0].AlignA, res[0].AlignB, nil
}

// NeedlemanWunsch is align.go:100-166 (the traceback stops when either index reaches 0, as the reference's does).
func NeedlemanWunsch(stringA string, stringB string, scoring Scoring) (int, string, string, error) {
	A, offA := polyhip.Pack([]string{stringA})
	B, _ := polyhip.Pack([]string{stringB})
	B = B[:len(stringB) : len(stringB)+1]
	raw, err := scoring.handle().NWAlignBatch(A, offA, B, nil, len(stringA))
	if err != nil {
		panic(err)
	}
	if raw[0].Err != 0 {
		return 0, "", "", symbolError(raw[0].Err)
	}
	return int(raw[0].Score), raw[0].AlignA, raw[0].AlignB, nil
}

// Alignment is one result of SmithWatermanBatch.
type Alignment struct {
	Score          int
	AlignA, AlignB string
	Err            error
}

// SmithWatermanBatch aligns every read against one shared reference in one device call.
func SmithWatermanBatch(reads []string, reference string, scoring *Scoring) []Alignment {
	A, offA := polyhip.Pack(reads)
	B, _ := polyhip.Pack([]string{reference})
	B = B[:len(reference):len(reference)+1]
	maxLen := 0
	for _, r := range reads {
		if len(r) > maxLen {
			maxLen = len(r)
		}
	}
	raw, err := scoring.handle().SWAlignBatch(A, offA, B, nil, maxLen)
	if err != nil {
		panic(err)
	}
	out := make([]Alignment, len(reads))
	for i, r := range raw {
		if r.Err != 0 {
			out[i].Err = symbolError(r.Err)
			continue
		}
		out[i] = Alignment{Score: int(r.Score), AlignA: r.AlignA, AlignB: r.AlignB}
	}
	return out
}
