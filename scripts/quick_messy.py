"""End-to-end on messy data: 100k reads (1000 families x 100 copies) of which some per cent are homopolymers,
tandem repeats or have a poly-A half -- K1 sketch time, K2 row-block time, against the clean set."""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import mash, bench_extra
dev = torch.device('cuda:0')
nfam, copies, L, k, s = 1000, 100, 10_000, 21, 1000
N = nfam * copies
def build(frac):
    g = torch.empty(nfam * L, dtype=torch.uint8, device=dev); mash.synth_dna_dev(0xC3, g)
    seqs = g.view(nfam, 1, L).expand(nfam, copies, L).contiguous().view(N, L)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for c0 in range(0, N, 20000):
        blk = seqs[c0:c0 + 20000]
        hit = torch.rand(blk.shape, device=dev, generator=gen) < 0.01
        blk[hit] = lut[torch.randint(0, 4, (int(hit.sum()),), device=dev, generator=gen)]
    if frac > 0:
        r = torch.rand(N, device=dev, generator=gen)
        seqs[r < frac] = ord("A")                                   # homopolymers
        unit = torch.tensor(list(b"ACGTTGCA" * 7), dtype=torch.uint8, device=dev)
        rep = unit.repeat(L // unit.numel() + 1)[:L]
        seqs[(r >= frac) & (r < 2 * frac)] = rep                    # tandem repeats
        tail = (r >= 2 * frac) & (r < 3 * frac)
        half = seqs[tail]; half[:, L // 2:] = ord("A"); seqs[tail] = half  # poly-A half
    return seqs.reshape(-1).contiguous()
def t(f, R=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R
offs = torch.arange(0, (N + 1) * L, L, dtype=torch.int64, device=dev)
for frac in (0.0, 0.001, 0.01):
    seqs = build(frac)
    sk = torch.zeros((N, s), dtype=torch.int32, device=dev)
    ms1 = t(lambda: mash.sketch_batch_dev(seqs, offs, k, s, sk))
    nrows = N // 8
    counts = torch.empty((nrows, N), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
    ms2 = t(lambda: mash.shared_counts_dev(sk[:nrows], sk, counts, work), 2)
    print(f"{3*frac:6.1%} low-complexity reads: sketch {ms1:8.2f} ms, 12.5k x 100k shared counts {ms2:9.2f} ms, mode {mash.shared_counts_mode(work)}", flush=True)
    del seqs, sk, counts, work
