import sys, torch
sys.path.insert(0, '.')
from poly_amd import bench_extra
print(bench_extra.fasta_feeder(torch.device('cuda:0')))
