// Drop-in overlay of the Tm scorers of github.com/bebop/poly/primers (primers.go:70-128) over libpolyhip.
//
// The fork keeps the reference's primers.go with two declaration renames (go/fork.sh): SantaLucia -> santaLuciaCPU,
// MarmurDoty -> marmurDotyCPU; MeltingTemp (:121-128), the thermodynamic tables and the barcode generator stay the
// reference's code.  One sequence is 30 dinucleotide lookups: the exported single-call functions run the reference's
// body (polyhip.MinTmCalls), the batch / scan entry points below are what uses the device.  UNCOMPILED here.
package primers

import (
	"math"

	"github.com/bebop/poly/internal/polyhip"
)

// SantaLucia is primers.go:70-105.  Bit-identical results (same fp64 operation order, Go's math.Log algorithm).
func SantaLucia(sequence string, primerConcentration, saltConcentration, magnesiumConcentration float64) (meltingTemp, dH, dS float64) {
	if 1 < polyhip.MinTmCalls {
		return santaLuciaCPU(sequence, primerConcentration, saltConcentration, magnesiumConcentration) // the reference's body
	}
	buf, offs := polyhip.Pack([]string{sequence})
	tm, h, s, err := polyhip.SantaLuciaBatch(buf, offs, primerConcentration, saltConcentration, magnesiumConcentration)
	if err != nil {
		panic(err)
	}
	return tm[0], h[0], s[0]
}

// MarmurDoty is primers.go:108-118.
func MarmurDoty(sequence string) float64 {
	if 1 < polyhip.MinTmCalls {
		return marmurDotyCPU(sequence) // the reference's body
	}
	buf, offs := polyhip.Pack([]string{sequence})
	tm, err := polyhip.MarmurDotyBatch(buf, offs)
	if err != nil {
		panic(err)
	}
	return tm[0]
}

// SantaLuciaBatch scores many primers in one device call (bit-identical to SantaLucia of each).
func SantaLuciaBatch(sequences []string, primerConcentration, saltConcentration, magnesiumConcentration float64) (meltingTemp, dH, dS []float64) {
	// a sequence with a byte >= 0x80 is upper-cased as UTF-8 by the reference (primers.go:71: an invalid byte becomes
	// the three bytes of U+FFFD, so even the length changes): those go through the reference's own body, the device
	// gets a one-letter stand-in whose results are overwritten
	packed, odd := sequences, []int(nil)
	for i, q := range sequences {
		if !polyhip.IsASCII(q) {
			if odd == nil {
				packed = append([]string(nil), sequences...)
			}
			packed[i] = "A"
			odd = append(odd, i)
		}
	}
	buf, offs := polyhip.Pack(packed)
	tm, h, s, err := polyhip.SantaLuciaBatch(buf, offs, primerConcentration, saltConcentration, magnesiumConcentration)
	if err != nil {
		panic(err)
	}
	for _, i := range odd {
		tm[i], h[i], s[i] = santaLuciaCPU(sequences[i], primerConcentration, saltConcentration, magnesiumConcentration)
	}
	return tm, h, s
}

// TmTable is the result of SantaLuciaScan: Tm/DH/DS[(L-MinLen)*Stride + start]; NaN where the window runs off the end.
type TmTable struct {
	MinLen, MaxLen, Stride int
	Tm, DH, DS             []float64
}

// SantaLuciaScan evaluates every window of every length minLen..maxLen of genome in one device call; the
// grow-until-Tm loops of primers/pcr (pcr.go:47-53, 94-101) become lookups into this table.
func SantaLuciaScan(genome string, minLen, maxLen int, primerConcentration, saltConcentration, magnesiumConcentration float64) TmTable {
	if !polyhip.IsASCII(genome) && minLen >= 1 && minLen <= maxLen && len(genome) >= minLen {
		// bytes >= 0x80 (see SantaLuciaBatch): every window through the reference's own body -- slow, and what the
		// reference returns
		ld := len(genome) - minLen + 1
		t := TmTable{MinLen: minLen, MaxLen: maxLen, Stride: ld, Tm: make([]float64, ld*(maxLen-minLen+1)),
			DH: make([]float64, ld*(maxLen-minLen+1)), DS: make([]float64, ld*(maxLen-minLen+1))}
		for l := minLen; l <= maxLen; l++ {
			for i := 0; i < ld; i++ {
				at := (l-minLen)*ld + i
				if i+l > len(genome) {
					t.Tm[at], t.DH[at], t.DS[at] = math.NaN(), math.NaN(), math.NaN()
					continue
				}
				t.Tm[at], t.DH[at], t.DS[at] = santaLuciaCPU(genome[i:i+l], primerConcentration, saltConcentration, magnesiumConcentration)
			}
		}
		return t
	}
	tm, h, s, ld, err := polyhip.SantaLuciaScan([]byte(genome), minLen, maxLen, primerConcentration, saltConcentration, magnesiumConcentration)
	if err != nil {
		panic(err)
	}
	return TmTable{MinLen: minLen, MaxLen: maxLen, Stride: ld, Tm: tm, DH: h, DS: s}
}

// FirstPrimerLengths is the grow loop of primers/pcr (pcr.go:47-53) for EVERY start of genome in one device call:
// Len[i] = the shortest primer genome[i:i+Len[i]] of minLen..maxLen nucleotides whose MeltingTemp is not below targetTm
// (0 if none), Tm[i] its melting temperature.
func FirstPrimerLengths(genome string, minLen, maxLen int, targetTm float64) (Len []uint16, Tm []float64) {
	l, t, err := polyhip.SantaLuciaScanFirst([]byte(genome), minLen, maxLen, 500e-9, 50e-3, 0.0, targetTm)
	if err != nil {
		panic(err)
	}
	return l, t
}
