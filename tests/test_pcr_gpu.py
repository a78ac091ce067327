"""primers/pcr primer design on the batched Tm kernel.
Mirrors primers/pcr/example_test.go:10-55 (exact primer strings at targetTm 55.0)."""
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

GENE = ("aataattacaccgagataacacatcatggataaaccgatactcaaagattctatgaagctatttgaggcacttggtacgatcaagtcgcgctcaatgtttggtggcttcg"
        "gacttttcgctgatgaaacgatgtttgcactggttgtgaatgatcaacttcacatacgagcagaccagcaaacttcatctaacttcgagaagcaagggctaaaaccgtacg"
        "tttataaaaagcgtggttttccagtcgttactaagtactacgcgatttccgacgacttgtgggaatccagtgaacgcttgatagaagtagcgaagaagtcgttagaacaag"
        "ccaatttggaaaaaaagcaacaggcaagtagtaagcccgacaggttgaaagacctgcctaacttacgactagcgactgaacgaatgcttaagaaagctggtataaaatcag"
        "ttgaacaacttgaagagaaaggtgcattgaatgcttacaaagcgatacgtgactctcactccgcaaaagtaagtattgagctactctgggctttagaaggagcgataaacg"
        "gcacgcactggagcgtcgttcctcaatctcgcagagaagagctggaaaatgcgctttcttaa")


def test_examples():
    from poly_amd import pcr
    # example_test.go:49-55
    assert pcr.DesignPrimers(GENE, 55.0) == ("AATAATTACACCGAGATAACACATCATGG", "TTAAGAAAGCGCATTTTCCAGC")
    # example_test.go:39-47
    assert pcr.DesignPrimersWithOverhangs(GENE, "TTATAGGTCTCATACT", "ATGAAGAGACCATATA", 55.0) == (
        "TTATAGGTCTCATACTAATAATTACACCGAGATAACACATCATGG", "TATATGGTCTCTTCATTTAAGAAAGCGCATTTTCCAGC")


def test_batch_matches_grow_loop_on_oracle():
    """every CDS-like slice of a synthetic genome: the lookup equals the reference's grow-until-Tm loop"""
    from poly_amd import pcr
    g = bytes(orc.synth_dna(0xC5, 40_000))
    genes = [g[i:i + 900] for i in range(0, 39_000, 700)]
    got = pcr.DesignPrimersBatch(genes, 58.0)
    for gene, (fwd, rev) in zip(genes, got):
        s = gene.upper()
        L = 15
        while orc.melting_temp(s[:L]) < 58.0:
            L += 1
        assert fwd.encode() == s[:L]
        L = 15
        while orc.melting_temp(orc.reverse_complement(s[len(s) - L:])) < 58.0:
            L += 1
        assert rev.encode() == orc.reverse_complement(s[len(s) - L:])


# ---- pcr.Simulate (pcr.go:74-203) ------------------------------------------------------------
PCR_FRAGMENT = "TTATAGGTCTCATACT" + GENE.upper() + "ATGAAGAGACCATATA"


def test_simulate_goldens():
    """pcr_test.go:14-95, example_test.go:57-66 -- same assertions as the reference's tests"""
    from poly_amd import pcr
    frags, err = pcr.Simulate([GENE], 55.0, False, ["TTATAGGTCTCATACTAATAATTACACCGAGATAACACATCATGG",
                                                  "TATATGGTCTCTTCATTTAAGAAAGCGCATTTTCCAGC"])
    assert err is None and frags == [PCR_FRAGMENT]
    frags, err = pcr.Simulate([GENE], 55.0, False, ["TATATGGTCTCTTCATTTAAGAAAGCGCATTTTCCAGC",
                                                  "TTATAGGTCTCATACTAATAATTACACCGAGATAACACATCATGG", "CTGCAGGTCGACTCTAG"])
    assert err is None and frags == [PCR_FRAGMENT]
    frags, _ = pcr.Simulate([GENE], 55.0, False, ["gatactcaaagattctatgaagctatttgaggcacttggtacg",
                                                "tatcgctttgtaagcattcaatgcacctttctcttcaagttg",
                                                "gtcgttcctcaatctcgcagagaagagctggaaaatg"])
    assert len(frags) == 1
    frags, _ = pcr.Simulate([GENE], 55.0, True, ["actctgggctttagaaggagcgataaacggc",
                                               "aagtgcctcaaatagcttcatagaatctttgagtatcgg"])
    assert frags[0] == ("ACTCTGGGCTTTAGAAGGAGCGATAAACGGCACGCACTGGAGCGTCGTTCCTCAATCTCGCAGAGAAGAGCTGGAAAATGCGCTTTCTTAAAATAATTACACC"
                        "GAGATAACACATCATGGATAAACCGATACTCAAAGATTCTATGAAGCTATTTGAGGCACTT")
    _, err = pcr.Simulate([GENE], 55.0, False, ["AATAATTACACCGAGATAACACATCATGG",
                                               "CCATGATGTGTTATCTCGGTGTAATTATTTTAAGAAAGCGCATTTTCCAGC"])
    assert err == "Concatemerization detected in PCR."
    assert pcr.Simulate([GENE], 55.0, False, ["ACGT"]) == (None, "Primers are too short.")


def test_simulate_matches_oracle_on_multiplex_reactions():
    """designed primer pairs for slices of a synthetic genome, mixed into multiplex reactions (linear and
    circular templates, a primer that never reaches the target, lower-case input): SimulateSimple equals
    the restated reference, fragment for fragment and in the same order."""
    import random
    from oracle import pcr_ref
    from poly_amd import pcr
    rng = random.Random(279)
    g = bytes(orc.synth_dna(0xC5 + 9, 6_000)).decode()
    for trial in range(12):
        template = g[trial * 400: trial * 400 + 1200]
        plist, cut = [], 0
        for _ in range(rng.randint(1, 3)):
            a = rng.randint(0, 700)
            b = a + rng.randint(120, 400)
            cut = (a + b) // 2
            fwd, rev = pcr.DesignPrimers(template[a:b], 57.0 + trial % 5)
            plist += [rng.choice(["", "GGTCTCA", "ttatag"]) + fwd, rev.lower() if trial % 2 else rev]
        if trial % 3 == 0:
            plist.append("ATATATATAT")  # stays below the target: rejected
        if trial % 4 == 0:  # origin inside the last amplicon: only a circular template amplifies it
            template = template[cut:] + template[:cut]
        for circular in (False, True):
            want = pcr_ref.simulate_simple([template, template.lower()], 55.0, circular, list(plist))
            got = pcr.SimulateSimple([template, template.lower()], 55.0, circular, list(plist))
            assert got == want, (trial, circular)
            w2, e2 = pcr_ref.simulate([template], 55.0, circular, list(plist))
            g2, e1 = pcr.Simulate([template], 55.0, circular, list(plist))
            assert (g2, e1) == (w2, e2), (trial, circular)
