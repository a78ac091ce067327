// Package align: drop-in for github.com/bebop/poly/search/align over libpolyhip: Scoring / NewScoring / Score
// (align.go:73-95), SmithWaterman (:171-232) and NeedlemanWunsch (:100-166) all run on the device.
// UNCOMPILED here (no Go toolchain in the authoring image).
package align

import (
	"sync"

	"github.com/bebop/poly/alphabet"
	"github.com/bebop/poly/internal/polyhip"
	"github.com/bebop/poly/search/align/matrix"
)

// Scoring is align.go:73-76: exactly the reference's two exported fields, so that struct literals and
// by-value copies (SmithWaterman takes a Scoring, not a *Scoring) keep working.
type Scoring struct {
	SubstitutionMatrix *matrix.SubstitutionMatrix
	GapPenalty         int
}

// Device-side scoring tables are cached per (matrix pointer, gap penalty) at package level: a Scoring is
// passed by value, so a handle stored inside it would land on a copy and every call would flatten the
// matrix again (16k Score() lookups + 7 device allocations).  The reference's canned matrices are
// package-level values that never change; a caller that mutates a matrix in place must call ForgetScoring.
type scoringKey struct {
	m   *matrix.SubstitutionMatrix
	gap int
}

var (
	scoringMu    sync.Mutex
	scoringCache = map[scoringKey]*polyhip.Scoring{}
)

const scoringCacheLimit = 64 // handles are tiny (a few KB of HBM each); bound the map anyway

// ForgetScoring drops the cached device tables of a matrix (all gap penalties).
func ForgetScoring(m *matrix.SubstitutionMatrix) {
	scoringMu.Lock()
	defer scoringMu.Unlock()
	for k := range scoringCache {
		if k.m == m {
			delete(scoringCache, k) // released by polyhip.Scoring's finalizer once no call uses it
		}
	}
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

// handle flattens through the matrix's PUBLIC Score(): its score table is unexported (matrix.go:13-17).
func (s Scoring) handle() *polyhip.Scoring {
	key := scoringKey{s.SubstitutionMatrix, s.GapPenalty}
	scoringMu.Lock()
	defer scoringMu.Unlock()
	if h, ok := scoringCache[key]; ok {
		return h
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
	if len(scoringCache) >= scoringCacheLimit {
		for k := range scoringCache { // drop an arbitrary entry; a call still using it keeps it alive, then
			delete(scoringCache, k) // polyhip.Scoring's finalizer releases the device tables
			break
		}
	}
	scoringCache[key] = h
	return h
}

func symbolError(code uint32) error {
	return &alphabet.Error{Message: "Symbol " + string(rune(code&0xFF)) + " not in alphabet"} // alphabet.go:38
}

// SmithWaterman is align.go:171-232.
func SmithWaterman(stringA string, stringB string, scoring Scoring) (int, string, string, error) {
	res := SmithWatermanBatch([]string{stringA}, stringB, scoring)
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
func SmithWatermanBatch(reads []string, reference string, scoring Scoring) []Alignment {
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
