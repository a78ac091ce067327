"""Times K4 (SantaLucia scan) at BASELINE config 5 size on one GPU."""
import sys
import torch
sys.path.insert(0, '.')
from poly_amd import primers, mash
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
Lmin, Lmax = 18, 30
g = torch.empty(n, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC5, g)
ns, nl = n - Lmin + 1, Lmax - Lmin + 1
out = [torch.zeros(nl * ns, dtype=torch.float64, device=dev) for _ in range(3)]
def step(a=Lmin, b=Lmax):
    primers.santalucia_scan_dev(g, n, 0, ns, a, b, 500e-9, 50e-3, 0.0, *out, ns)
for lo, hi, tag in ((18, 30, "fixed<18,30>"), (17, 30, "generic 17..30")):
    nl2 = hi - lo + 1
    if nl2 * (n - lo + 1) > out[0].numel():
        out = [torch.zeros(nl2 * (n - lo + 1), dtype=torch.float64, device=dev) for _ in range(3)]
    def st():
        primers.santalucia_scan_dev(g, n, 0, n - lo + 1, lo, hi, 500e-9, 50e-3, 0.0, *out, n - lo + 1)
    st(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 10
    e0.record()
    for _ in range(R):
        st()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    win = sum(n - L + 1 for L in range(lo, hi + 1))
    print(f"K4 {tag}: {ms:.3f} ms per {n} B genome -> {win/ms*1e3:.3e} windows/s, {win*24.1/ms*1e3/1e9:.1f} GB/s algorithmic")
