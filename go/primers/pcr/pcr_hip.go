// Package pcr: the Tm-bound parts of github.com/bebop/poly/primers/pcr on the batched device Tm scorer.
//
// DesignPrimersWithOverhangs / DesignPrimers (pcr.go:44-66) keep their signatures and, for one gene, the reference's
// body (go/fork.sh renames the former to designPrimersWithOverhangsCPU; DesignPrimers stays the reference's and calls
// the overlay).  DesignPrimersBatch scores the grow-until-Tm loops (:47-53) of many genes with ONE
// polyhip_santalucia_batch call per growth window ("design primers for every
// CDS of a genome", tutorials/002_primer_design_test.go:82-99).  MinimalBindingLengths replaces the per-primer
// loop of SimulateSimple (pcr.go:95-101); the rest of SimulateSimple / Simulate (site lookup through
// index/suffixarray, fragment assembly; :106-203) is host orchestration that stays the reference's code and
// calls MinimalBindingLengths once per reaction instead of MeltingTemp once per suffix.  UNCOMPILED here.
package pcr

import (
	"strings"

	"github.com/bebop/poly/internal/polyhip"
	"github.com/bebop/poly/transform"
)

// minimalPrimerLength (7, pcr.go:35) and designedMinimalPrimerLength (15, pcr.go:38) are the reference's constants.

// primers.MeltingTemp's conditions (primers.go:122-124)
const (
	primerConcentration    = 500e-9
	saltConcentration      = 50e-3
	magnesiumConcentration = 0.0
)

func meltingTemps(candidates []string) []float64 {
	buf, offs := polyhip.Pack(candidates)
	tm, _, _, err := polyhip.SantaLuciaBatch(buf, offs, primerConcentration, saltConcentration, magnesiumConcentration)
	if err != nil {
		panic(err)
	}
	return tm
}

// PrimerPair is one result of DesignPrimersBatch.
type PrimerPair struct{ Forward, Reverse string }

// DesignPrimersBatch returns, for every sequence, the shortest prefix and the shortest reverse-complemented
// suffix of at least 15 nt whose MeltingTemp reaches targetTm -- what DesignPrimers returns for each, from one
// device call per growth window (windows of 64 lengths, quadrupling while some primer has not reached the
// target).  Like the reference, it panics (slice bounds) if a primer would have to outgrow its sequence.
func DesignPrimersBatch(sequences []string, targetTm float64) []PrimerPair {
	upper := make([]string, len(sequences))
	for i, s := range sequences {
		upper[i] = strings.ToUpper(s) // pcr.go:45
		_ = upper[i][0:designedMinimalPrimerLength]
	}
	out := make([]PrimerPair, len(sequences))
	pending := make([]int, len(sequences))
	for i := range pending {
		pending[i] = i
	}
	lo := designedMinimalPrimerLength
	for growth := 64; len(pending) > 0; growth *= 4 {
		type owner struct{ seq, strand, length int }
		var cands []string
		var owners []owner
		for _, i := range pending {
			s := upper[i]
			for L := lo; L < lo+growth && L <= len(s); L++ {
				if out[i].Forward == "" {
					cands = append(cands, s[:L])
					owners = append(owners, owner{i, 0, L})
				}
				if out[i].Reverse == "" {
					cands = append(cands, transform.ReverseComplement(s[len(s)-L:]))
					owners = append(owners, owner{i, 1, L})
				}
			}
		}
		tm := meltingTemps(cands)
		for j, o := range owners { // candidates of one (sequence, strand) are in increasing length
			if !(tm[j] < targetTm) {
				if o.strand == 0 && out[o.seq].Forward == "" {
					out[o.seq].Forward = cands[j]
				} else if o.strand == 1 && out[o.seq].Reverse == "" {
					out[o.seq].Reverse = cands[j]
				}
			}
		}
		next := pending[:0]
		for _, i := range pending {
			if out[i].Forward == "" || out[i].Reverse == "" {
				if lo+growth > len(upper[i]) {
					panic("slice bounds out of range (no primer of this sequence reaches the target Tm, pcr.go:48)")
				}
				next = append(next, i)
			}
		}
		pending = next
		lo += growth
	}
	return out
}

// DesignPrimersWithOverhangs is pcr.go:44-60 (renamed designPrimersWithOverhangsCPU in the fork's pcr.go; one gene is
// two grow loops of a few MeltingTemp calls each -- the reference's body; DesignPrimersBatch is the device entry point).
func DesignPrimersWithOverhangs(sequence, forwardOverhang, reverseOverhang string, targetTm float64) (string, string) {
	return designPrimersWithOverhangsCPU(sequence, forwardOverhang, reverseOverhang, targetTm)
}

// MinimalBindingLengths is the loop of pcr.go:95-101 for every primer of a reaction: minimalLength[i] = the
// LAST suffix length (counting up from 7) whose MeltingTemp is still below targetTm -- 0 if the 7-mer already
// reaches it, len(primer) if the whole primer stays below (SimulateSimple then skips it, :104).  Suffixes are
// scored in growing windows (7..70, then x4), so a 100 kb amplicon used as a primer in Simulate's second round
// (pcr.go:181) costs the few dozen suffixes the reference would have scored, not all of them.
func MinimalBindingLengths(primerList []string, targetTm float64) []int {
	minimal := make([]int, len(primerList))
	done := make([]bool, len(primerList))
	lo := minimalPrimerLength
	for span := 64; ; span *= 4 {
		var cands []string
		var owners [][2]int
		for i, primer := range primerList {
			_ = primer[len(primer)-minimalPrimerLength:] // the reference's slice panics on a shorter primer
			if done[i] {
				continue
			}
			for index := lo; index < lo+span && index <= len(primer); index++ {
				cands = append(cands, primer[len(primer)-index:])
				owners = append(owners, [2]int{i, index})
			}
		}
		if len(cands) == 0 {
			return minimal
		}
		tm := meltingTemps(cands)
		for j, o := range owners {
			i, index := o[0], o[1]
			if done[i] {
				continue
			}
			if !(tm[j] < targetTm) {
				done[i] = true
				continue
			}
			minimal[i] = index
			if index == len(primerList[i]) {
				done[i] = true
			}
		}
		lo += span
	}
}
