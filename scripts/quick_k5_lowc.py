"""K5 on inputs that defeat candidate elimination: homopolymers / short periods (serial fallback per sequence)."""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import mash, seqhash
dev = torch.device('cuda:0')
n, L = 100_000, 5000
def run(name, seqs):
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    rot = torch.zeros(n, dtype=torch.int64, device=dev); out = torch.empty(n * L, dtype=torch.uint8, device=dev)
    seqhash.least_rotation_batch_dev(seqs, offs, L, rot, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): seqhash.least_rotation_batch_dev(seqs, offs, L, rot, out)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1)/3:8.3f} ms per {n} sequences")
rnd = torch.empty(n * L, dtype=torch.uint8, device=dev); mash.synth_dna_dev(1, rnd)
run("random DNA", rnd)
run("homopolymer", torch.full((n * L,), ord("A"), dtype=torch.uint8, device=dev))
unit = torch.tensor(list(b"ACGTTGCA" * 7), dtype=torch.uint8, device=dev)
run("tandem repeat (period 56)", unit.repeat(n * L // unit.numel() + 1)[: n * L].contiguous())
half = rnd.clone().view(n, L); half[:, L // 2:] = ord("A")
run("half random, half poly-A", half.view(-1).contiguous())
