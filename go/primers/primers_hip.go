// Drop-in overlay of the Tm scorers of github.com/bebop/poly/primers (primers.go:70-128) over libpolyhip.
//
// The fork keeps the reference's primers.go with two declaration renames (go/fork.sh): SantaLucia -> santaLuciaCPU,
// MarmurDoty -> marmurDotyCPU; MeltingTemp (:121-128), the thermodynamic tables and the barcode generator stay the
// reference's code.  One sequence is 30 dinucleotide lookups: the exported single-call functions run the reference's
// body (polyhip.MinTmCalls), the batch / scan entry points below are what uses the device.  UNCOMPILED here.
package primers

import "github.com/bebop/poly/internal/polyhip"

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
	buf, offs := polyhip.Pack(sequences)
	tm, h, s, err := polyhip.SantaLuciaBatch(buf, offs, primerConcentration, saltConcentration, magnesiumConcentration)
	if err != nil {
		panic(err)
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
