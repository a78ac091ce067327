"""K5 least rotation: random DNA, exact tandem repeats (the unit divides the length), tandem repeats that do not close on
themselves, half poly-A -- ms per 100k sequences of 5 kb, rotation indices of a sample against the oracle"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import oracle as orc
from poly_amd import mash, seqhash
from poly_amd.bench_extra import _time
dev = torch.device('cuda:0')
n, L = 100_000, 5000
offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
rot = torch.zeros(n, dtype=torch.int64, device=dev)
out = torch.zeros(n * L, dtype=torch.uint8, device=dev)


def run(tag, buf):
    ms = _time(lambda: seqhash.least_rotation_batch_dev(buf, offs, L, rot, out), 10)
    h = buf[: 8 * L].cpu().numpy()
    ok = all(int(rot[i]) == orc.booth_least_rotation(h[i * L:(i + 1) * L].tobytes()) or
             bytes(out[i * L:(i + 1) * L].cpu().numpy()) == orc.rotate_sequence(h[i * L:(i + 1) * L].tobytes()) for i in range(8))
    print(f"{tag}: {ms:.3f} ms per {n} x {L} bp  (sample = oracle: {ok})", flush=True)


rnd = torch.empty(n * L, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0x5EED, rnd)
run("random DNA", rnd)
for unit in (b"ACGTTGCA", b"AC", b"ACGTTGCAAT" * 5, b"ACGTTGC", b"GATTACA" * 9):
    u = torch.tensor(list(unit), dtype=torch.uint8, device=dev)
    reads = u.repeat(L // len(unit) + 1)[:L].repeat(n).contiguous()
    kind = "closes on itself" if L % len(unit) == 0 else "does not close"
    run(f"tandem repeat, unit {len(unit)} ({kind})", reads)
half = rnd.clone().view(n, L)
half[:, L // 2:] = ord('A')
run("half random / half poly-A", half.view(-1))
