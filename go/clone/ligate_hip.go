// Drop-in overlay of clone.CircularLigate (github.com/bebop/poly/clone, clone.go:270-340) on the batched seqhash kernel.
//
// The fork keeps the reference's clone.go with one declaration rename (go/fork.sh): CircularLigate -> circularLigateCPU
// (recurseLigate, Fragment, GoldenGate ... stay the reference's code; GoldenGate calls the CircularLigate below).
//
// recurseLigate calls seqhash.Hash once per candidate construct (clone.go:275 for a circularised construct, :305 for an
// "infinite" linear one) and keeps the construct only if the hash is new.  No hash steers the recursion -- both call
// sites are leaves that return from the frame whatever the map says -- so the recursion runs twice: a DRY pass that only
// records the Hash calls, ONE seqhash.HashBatch per flag group for all of them, and the real pass fed from those
// results: exactly the constructs, in exactly the order, the reference keeps (its quirks included: an endless ligation
// returns from the whole frame; usedFragments grows across siblings because :314 re-assigns the frame's own slice).
// Same driver as poly_amd/clone.py, which the tests run against clone/example_test.go:11-31 and clone_test.go:142-214.
// UNCOMPILED in the authoring image.
package clone

import (
	"github.com/bebop/poly/internal/polyhip"
	"github.com/bebop/poly/seqhash"
	"github.com/bebop/poly/transform"
)

type hashCall struct {
	construct string
	circular  bool
}

// ligate is recurseLigate (clone.go:270-323) with the Hash call behind hashOf.
func ligate(seed Fragment, pool []Fragment, used []Fragment, seen map[string]struct{}, hashOf func(string, bool) string) (open []string, infinite []string) {
	if seed.ForwardOverhang == seed.ReverseOverhang { // :273
		construct := seed.ForwardOverhang + seed.Sequence
		h := hashOf(construct, true)
		if _, ok := seen[h]; ok {
			return nil, nil
		}
		seen[h] = struct{}{}
		return []string{construct}, nil
	}
	for _, next := range pool { // :284
		var newSeed Fragment
		attached := false
		if seed.ReverseOverhang == next.ForwardOverhang { // :288
			attached = true
			newSeed = Fragment{seed.Sequence + seed.ReverseOverhang + next.Sequence, seed.ForwardOverhang, next.ReverseOverhang}
		}
		if seed.ReverseOverhang == transform.ReverseComplement(next.ReverseOverhang) &&
			seed.ReverseOverhang != transform.ReverseComplement(seed.ReverseOverhang) { // :294
			attached = true
			newSeed = Fragment{seed.Sequence + seed.ReverseOverhang + transform.ReverseComplement(next.Sequence), seed.ForwardOverhang,
				transform.ReverseComplement(next.ForwardOverhang)}
		}
		if !attached {
			continue
		}
		for _, u := range used { // :302
			if u.Sequence == next.Sequence {
				construct := u.ForwardOverhang + u.Sequence + u.ReverseOverhang
				h := hashOf(construct, false)
				if _, ok := seen[h]; ok {
					return nil, nil
				}
				seen[h] = struct{}{}
				return nil, []string{construct}
			}
		}
		used = append(used, next) // :314
		o, i := ligate(newSeed, pool, used, seen, hashOf)
		open = append(open, o...)
		infinite = append(infinite, i...)
	}
	return open, infinite
}

// CircularLigate is clone.go:326-340.
func CircularLigate(fragments []Fragment) ([]string, []string) {
	// dry pass: which constructs get hashed, in which order (every call answers a never-seen value)
	var calls []hashCall
	serial := 0
	record := func(construct string, circular bool) string {
		calls = append(calls, hashCall{construct, circular})
		serial++
		return string(rune(serial)) + "#" + construct // unique per call: the dry pass takes the "new hash" branch everywhere
	}
	for _, f := range fragments {
		ligate(f, fragments, []Fragment{}, map[string]struct{}{}, record)
	}
	if len(calls) < polyhip.MinLigateHashes {
		return circularLigateCPU(fragments) // the reference's body: a handful of Hash calls
	}
	// one device call per flag group
	var circ, lin []string
	for _, c := range calls {
		if c.circular {
			circ = append(circ, c.construct)
		} else {
			lin = append(lin, c.construct)
		}
	}
	var hc, hl []string
	if len(circ) > 0 {
		hc, _ = seqhash.HashBatch(circ, seqhash.DNA, true, true) // clone.go:275 drops Hash's error: "" on a bad letter
	}
	if len(lin) > 0 {
		hl, _ = seqhash.HashBatch(lin, seqhash.DNA, false, true)
	}
	// real pass, fed from the batch
	at, ic, il := 0, 0, 0
	replay := func(construct string, circular bool) string {
		if at >= len(calls) || calls[at].construct != construct || calls[at].circular != circular {
			panic("clone.CircularLigate: the replay diverged from the dry pass")
		}
		at++
		if circular {
			ic++
			return hc[ic-1]
		}
		il++
		return hl[il-1]
	}
	var out, inf []string
	seen := map[string]struct{}{}
	for _, f := range fragments {
		o, i := ligate(f, fragments, []Fragment{}, seen, replay)
		out = append(out, o...)
		inf = append(inf, i...)
	}
	return out, inf
}
