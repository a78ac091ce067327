// Drop-in overlay of github.com/bebop/poly/search/align over libpolyhip.
//
// The fork keeps the reference's align.go with two declaration renames (go/fork.sh): SmithWaterman -> smithWatermanCPU,
// NeedlemanWunsch -> needlemanWunschCPU.  type Scoring, NewScoring and Scoring.Score (align.go:73-95) stay the
// reference's code; this file supplies the two exported functions again -- small inputs run the reference's body, large
// ones the device -- and SmithWatermanBatch.  UNCOMPILED here (no Go toolchain in the authoring image).
package align

import (
	"sync"

	"github.com/bebop/poly/alphabet"
	"github.com/bebop/poly/internal/polyhip"
	"github.com/bebop/poly/search/align/matrix"
)

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

// handle flattens through the matrix's PUBLIC Score(): its score table is unexported (matrix.go:13-17).
func handle(s Scoring) *polyhip.Scoring {
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
	if len(stringA)*len(stringB) < polyhip.MinAlignCells {
		return smithWatermanCPU(stringA, stringB, scoring) // the reference's body
	}
	res := SmithWatermanBatch([]string{stringA}, stringB, scoring)
	if res[0].Err != nil {
		return 0, "", "", res[0].Err
	}
	return res[0].Score, res[0].AlignA, res[0].AlignB, nil
}

// NeedlemanWunsch is align.go:100-166 (the traceback stops when either index reaches 0, as the reference's does).
func NeedlemanWunsch(stringA string, stringB string, scoring Scoring) (int, string, string, error) {
	if len(stringA)*len(stringB) < polyhip.MinAlignCells {
		return needlemanWunschCPU(stringA, stringB, scoring) // the reference's body
	}
	A, offA := polyhip.Pack([]string{stringA})
	B, _ := polyhip.Pack([]string{stringB})
	B = B[:len(stringB) : len(stringB)+1]
	raw, err := handle(scoring).NWAlignBatch(A, offA, B, nil, len(stringA))
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
	raw, err := handle(scoring).SWAlignBatch(A, offA, B, nil, maxLen)
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
