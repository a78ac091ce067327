"""How many pairs the packed SW pass hands to the exact kernel, and why (tie flag vs. cell not found)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from poly_amd import align, alphabet, matrix, mash
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
LA, LB = 150, 5000
a = alphabet.NewAlphabet(list("-ACGT"))
sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
B = torch.empty(LB, dtype=torch.uint8, device=dev)
mash.synth_dna_dev(0xC4, B)
gen = torch.Generator(device=dev); gen.manual_seed(0xC4)
starts = torch.randint(0, LB - LA, (n,), device=dev, generator=gen)
A = B[starts[:, None] + torch.arange(LA, device=dev)[None, :]]
hit = torch.rand(A.shape, device=dev, generator=gen) < 0.05
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
A[hit] = lut[torch.randint(0, 4, (int(hit.sum()),), device=dev, generator=gen)]
A = A.reshape(-1).contiguous()
offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
score = torch.zeros(n, dtype=torch.int64, device=dev)
ea, eb, er = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
work = torch.zeros(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work)
torch.cuda.synchronize()
w = work.cpu().numpy()
fast = 256 + (LB * 8 + 255) // 256 * 256
cnt = int(w[fast:fast + 4].view(np.uint32)[0])
prof2 = (LB // 4 * 36 * 16 + 255) // 256 * 256
info = (n * 4 + 255) // 256 * 256
infoM = w[fast + 256 + prof2: fast + 256 + prof2 + 4 * n].view(np.uint32)
infoQ = w[fast + 256 + prof2 + info: fast + 256 + prof2 + info + 4 * n].view(np.uint32)
lst = w[fast + 256 + prof2 + 2 * info: fast + 256 + prof2 + 2 * info + 4 * cnt].view(np.uint32)
tie = (infoQ >> 31).astype(bool)
print(f"path {align.last_path()} pairs {n} listed {cnt} ({cnt/n:.2%}); tie-flagged {int(tie.sum())}; listed but not tie-flagged {int((~tie[lst]).sum())}")
s = score.cpu().numpy(); eb_h = eb.cpu().numpy(); ea_h = ea.cpu().numpy()
for p in lst[:8]:
    print("pair", p, "M", infoM[p], "q", infoQ[p] & 0x7FFFFFFF, "tie", tie[p], "-> exact: score", s[p], "endA", ea_h[p], "endB", eb_h[p], "block of endB", (eb_h[p] - 1) // 4)
