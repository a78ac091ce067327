"""K1 on inputs that defeat the threshold pass: tandem repeats / homopolymers (every read goes to the general kernel)."""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import mash
dev = torch.device('cuda:0')
n, L, k, s = 100_000, 10_000, 21, 1000
def run(name, seqs):
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    out = torch.zeros((n, s), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(seqs, offs, k, s, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): mash.sketch_batch_dev(seqs, offs, k, s, out)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1)/3:8.3f} ms per {n} reads")
rnd = torch.empty(n * L, dtype=torch.uint8, device=dev); mash.synth_dna_dev(1, rnd)
run("random DNA", rnd)
unit = torch.tensor(list(b"ACGTTGCA" * 7), dtype=torch.uint8, device=dev)  # period 56 -> 56 distinct k-mers
run("tandem repeat (period 56)", unit.repeat(n * L // unit.numel() + 1)[: n * L].contiguous())
run("homopolymer", torch.full((n * L,), ord("A"), dtype=torch.uint8, device=dev))
half = rnd.clone().view(n, L); half[:, L // 2:] = ord("A")
run("half random, half poly-A", half.view(-1).contiguous())
