"""Times K2 (all-vs-all shared counts + distance) on a C3-like set: families of mutated copies."""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import mash
dev = torch.device('cuda:0')
nfam = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
copies, L, k, s = 100, 10_000, 21, 1000
N = nfam * copies
g = torch.empty(nfam * L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC3, g)
seqs = g.view(nfam, 1, L).expand(nfam, copies, L).contiguous().view(N, L)
gen = torch.Generator(device=dev); gen.manual_seed(0xC3)
for c0 in range(0, N, 10_000):
    blk = seqs[c0:c0 + 10_000]
    hit = torch.rand(blk.shape, device=dev, generator=gen) < 0.01
    rnd = torch.randint(0, 4, blk.shape, device=dev, generator=gen, dtype=torch.uint8)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    blk[hit] = lut[rnd[hit].long()]
offs = torch.arange(0, (N + 1) * L, L, dtype=torch.int64, device=dev)
sk = torch.zeros((N, s), dtype=torch.int32, device=dev)
mash.sketch_batch_dev(seqs.view(-1), offs, k, s, sk)
torch.cuda.synchronize()
del seqs, g
for nrows, tag in ((N // 8, "1/8 row block (one rank of 8)"), (N, "full matrix on one GPU")):
    X = sk[:nrows]
    counts = torch.empty((nrows, N), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
    mash.shared_counts_dev(X, sk, counts, work); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 3
    e0.record()
    for _ in range(R):
        mash.shared_counts_dev(X, sk, counts, work)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    mode = mash.shared_counts_mode(work)
    pairs = nrows * N
    nz = int((counts != 0).sum()); tot = int(counts.to(torch.int64).sum())
    print(f"K2 {tag}: {ms:.3f} ms for {nrows}x{N} pairs -> {pairs/ms*1e3:.3e} pairs/s, {pairs*2/ms*1e3/1e9:.1f} GB/s (u16 out); "
          f"mode={mode} nonzero={nz} shared_total={tot} diag_ok={bool((counts[:, :nrows].diagonal() == s).all())}")
    if nrows == N // 8:
        dist = torch.empty((nrows, N), dtype=torch.float64, device=dev)
        mash.distance_from_counts_dev(counts, s, s, dist); torch.cuda.synchronize()
        e0.record()
        for _ in range(R):
            mash.distance_from_counts_dev(counts, s, s, dist)
        e1.record(); torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / R
        print(f"   distance_from_counts: {ms2:.3f} ms -> {pairs*10/ms2*1e3/1e9:.1f} GB/s (2 B in + 8 B out)")
        del dist
    del counts, work
