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
