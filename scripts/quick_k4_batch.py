"""polyhip_santalucia_batch_dev / marmurdoty_batch_dev: n primers of 18..30 bp, device-resident."""
import sys, torch
sys.path.insert(0, '.')
from poly_amd import _lib, mash
dev = torch.device('cuda:0')
n = 5_000_000
gen = torch.Generator(device=dev); gen.manual_seed(1)
lens = torch.randint(18, 31, (n,), device=dev, generator=gen, dtype=torch.int64)
offs = torch.zeros(n + 1, dtype=torch.int64, device=dev); offs[1:] = torch.cumsum(lens, 0)
tot = int(offs[-1])
seqs = torch.empty(tot, dtype=torch.uint8, device=dev); mash.synth_dna_dev(9, seqs)
tm, dh, ds = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
L = _lib.lib()
def t(f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); f(); f(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
ms = t(lambda: _lib.check(L.polyhip_santalucia_batch_dev(seqs.data_ptr(), offs.data_ptr(), n, 500e-9, 50e-3, 0.0, tm.data_ptr(), dh.data_ptr(), ds.data_ptr(), None)))
print(f"santalucia_batch: {ms:.3f} ms per {n} primers -> {n/ms*1e3:.3e} primers/s ({(tot + 8*n + 24*n)/ms*1e3/1e9:.0f} GB/s algorithmic)")
ms = t(lambda: _lib.check(L.polyhip_marmurdoty_batch_dev(seqs.data_ptr(), offs.data_ptr(), n, tm.data_ptr(), None)))
print(f"marmurdoty_batch: {ms:.3f} ms per {n} primers -> {n/ms*1e3:.3e} primers/s")
