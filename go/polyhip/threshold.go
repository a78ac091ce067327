package polyhip

import (
	"os"
	"strconv"
)

// Size thresholds of the drop-in (SURVEY.md 8b: "single-call functions keep reference behaviour -- CPU path below a
// size threshold").  A device call costs pack + H2D + launch + D2H + sync, a few hundred microseconds whatever the
// size; the reference's own Go body (kept in the fork as *CPU functions, see go/fork.sh) finishes a 62-mer Sketch or a
// GTTGAC x GTTAC alignment in less than that.  Below the threshold the exported function runs the reference's body,
// above it the batch entry point -- so the reference's own unit tests run at their old speed through the drop-in and
// BASELINE configs[0] (phiX174, "plumbing, no GPU") stays a CPU run.  These are NOT a fallback for a missing device:
// above the threshold a failing device call panics.
//
// Defaults are the measured break-even of the Python host layer on one MI355X (DESIGN.md section 1, scripts/quick_single.py),
// rounded up; POLYHIP_GO_MIN_WORK=<n> scales all of them (0 = always use the device: what the parity tests set).
var (
	MinSketchBytes    = 1 << 16 // mash.Sketch: sequence bytes (phiX174's 5,386 bp stay on the CPU)
	MinAlignCells     = 1 << 22 // align.SmithWaterman / NeedlemanWunsch: len(A) * len(B)
	MinTmCalls        = 1 << 30 // primers.SantaLucia / MarmurDoty on ONE sequence: always the CPU (use the batch / scan API)
	MinRotateBytes    = 1 << 16 // seqhash.RotateSequence / Hash: sequence bytes
	MinLigateHashes   = 64      // clone.CircularLigate: candidate constructs to hash
	MinDistancePairs  = 1 << 12 // mash Similarity / Distance of ONE pair: always the CPU merge; DistanceMatrix above this
)

func init() {
	if v, err := strconv.Atoi(os.Getenv("POLYHIP_GO_MIN_WORK")); err == nil && v == 0 {
		MinSketchBytes, MinAlignCells, MinTmCalls, MinRotateBytes, MinLigateHashes, MinDistancePairs = 0, 0, 0, 0, 0, 0
	}
}
