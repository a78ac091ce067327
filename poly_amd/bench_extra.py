"""Secondary rates reported next to bench.py's headline k-mers/s (BASELINE configs 3-5
and seqhash): SW cell updates/s, Tm windows/s, distance pairs/s, least-rotation bases/s.
Device-resident timing with events on the launch stream; synthetic inputs per SURVEY 8d."""
from __future__ import annotations

import torch

from . import align, alphabet, fasta, fastq, mash, matrix, primers, seqhash, workloads

HBM_PEAK_GBS = 8000.0


def _time(fn, reps: int = 10, warm: int = 5) -> float:
    """median of `reps` (>= 10) separately event-timed launches after `warm` (>= 5) untimed ones
    (>= 5 warm-ups, median of >= 10), in ms; events on the launch stream"""
    reps, warm = max(reps, 10), max(warm, 5)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return 0.5 * (ts[(reps - 1) // 2] + ts[reps // 2])


def _wall(fn, reps: int = 5, warm: int = 2) -> float:
    """host-pointer (PCIe-inclusive) calls are synchronous: median wall time in ms"""
    import time
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def family_sketches(dev, nfam: int, copies: int, L: int, k: int, s: int, seed: int, sub: float = 0.01):
    """`nfam` random genomes x `copies` copies at `sub` substitution rate, sketched by K1."""
    N = nfam * copies
    g = torch.empty(nfam * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(seed, g)
    seqs = g.view(nfam, 1, L).expand(nfam, copies, L).contiguous().view(N, L)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    step = max(1, min(N, (256 << 20) // L))
    for c0 in range(0, N, step):
        blk = seqs[c0:c0 + step]
        hit = torch.rand(blk.shape, device=dev, generator=gen) < sub
        rnd = torch.randint(0, 4, (int(hit.sum()),), device=dev, generator=gen)
        blk[hit] = lut[rnd]
    offs = torch.arange(0, (N + 1) * L, L, dtype=torch.int64, device=dev)
    sk = torch.zeros((N, s), dtype=torch.int32, device=dev)
    mash.sketch_batch_dev(seqs.view(-1), offs, k, s, sk)
    torch.cuda.synchronize()
    return sk


def sw(dev, n: int = 1_000_000, LA: int = 150, LB: int = 5000, shard: int = 0):
    """config 4: n reads of LA bp (windows of the reference, 5 % substitutions + 1 % indels) vs one LB reference, NUC_4, gap -2;
    `shard` picks this rank's reads when the pairs are split over GPUs (the reference is the same everywhere)"""
    a = alphabet.NewAlphabet(list("-ACGT"))
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
    # windows of the reference with 5 % substitutions + 1 % indels (poly_amd/workloads.py); rank
    # `shard` takes reads shard*n .. of the one definition
    B, A = workloads.config4_reads(n, LA, LB, first=shard * n, device=dev)
    A = A.reshape(-1).contiguous()
    offA = torch.arange(0, (n + 1) * LA, LA, dtype=torch.int64, device=dev)
    score = torch.zeros(n, dtype=torch.int64, device=dev)
    ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
    work = torch.empty(align.sw_workspace_bytes(sc, n, LA, LB, True), dtype=torch.uint8, device=dev)
    ms_score = _time(lambda: align.sw_batch_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, work), 3)
    half = align.last_packed_half()  # gfx950's half-float cell (scores < 2048): 17 instructions per row and block, else 22
    # VALU-issue ceiling: the kernel's own instruction count per row and 4-column block (counter-measured for the
    # half-float cell, profiles/k3_issue.json; every instruction of the packed recurrence is a 4-cycle one,
    # profiles/r03_valu_mix.json) at the clock the kernel sustains; beside it the ceiling of the 14-instruction recurrence
    # floor, which says how far the stream is from the algorithm, not from its own issue rate
    import json as _json
    import os as _os
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    try:
        _k3 = _json.load(open(_os.path.join(_root, "profiles", "k3_issue.json")))
    except Exception:
        _k3 = {"valu_instructions_per_row_block": 17.15, "clock_GHz": 2.343, "floor_instructions_per_row_block": 14, "source": "defaults"}
    per_blk = _k3["valu_instructions_per_row_block"] if half else 22
    clock = _k3["clock_GHz"] * 1e9
    ceiling = 1024 * clock * 512 / (4 * per_blk)
    ceiling_floor = 1024 * clock * 512 / (4 * _k3["floor_instructions_per_row_block"])
    stride = align.sw_traceback_stride(sc, LA, LB)
    tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, LA, LB), dtype=torch.uint8, device=dev)
    alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    ms_tb = _time(lambda: align.sw_traceback_dev(sc, A, offA, LA, B, None, LB, ea, eb, er, alnA, alnB, ln, tbw, score_t=score), 2)
    ms_fused = _time(lambda: align.sw_align_dev(sc, A, offA, LA, B, None, LB, score, ea, eb, er, alnA, alnB, ln, work, tbw), 2)
    cells = n * LA * LB
    alg = n * (LA + 8 + 8)  # read + score + end position per pair
    if (n, LA, LB) != (1_000_000, 150, 5000):  # other read lengths: the plain figures (which kernels ran is the library's choice)
        return {
            "workload": f"{n} x {LA} bp reads vs one {LB} bp reference, NUC_4, gap -2",
            "cell_updates_per_s": cells / ms_score * 1e3, "score_pass_ms": ms_score,
            "cell_updates_per_s_with_traceback": cells / (ms_score + ms_tb) * 1e3, "traceback_ms": ms_tb,
            "align_one_call_ms": ms_fused, "cell_updates_per_s_align_one_call": cells / ms_fused * 1e3,
            "score_path": align.last_path(), "traceback_path": align.sw_traceback_last_path(),
            "mean_score": float(score.double().mean()), "mean_alignment_len": float(ln.double().mean()),
        }
    return {
        "workload": f"{n} x {LA} bp reads vs one {LB} bp reference, NUC_4, gap -2 (BASELINE configs[3])",
        "cell_updates_per_s": cells / ms_score * 1e3, "score_pass_ms": ms_score,
        "cell_updates_per_s_with_traceback": cells / (ms_score + ms_tb) * 1e3, "traceback_ms": ms_tb,
        # polyhip_sw_align_batch_dev: score + strings in one call, end cell found by the traceback kernel
        "align_one_call_ms": ms_fused, "cell_updates_per_s_align_one_call": cells / ms_fused * 1e3,
        "algorithmic_GBs_score_pass": alg / ms_score * 1e3 / 1e9,
        # the bound that matters for K3: VALU issue.  The packed kernel spends `per_blk` instructions of four
        # issue cycles per 512 cells (64 lanes x 2 pairs x 4 columns) on 1024 SIMDs at the clock it sustains.
        "packed_cell": "half-float (v_pk_maximum3_f16)" if half else "int16",
        "valu_issue_ceiling_cell_updates_per_s": ceiling,
        "frac_of_valu_issue_ceiling": cells / ms_score * 1e3 / ceiling,
        "roofline": {"bound": "valu", "achieved": cells / ms_score * 1e3 / 1e12, "peak": ceiling / 1e12,
                     "unit": "T cell updates/s", "frac": cells / ms_score * 1e3 / ceiling,
                     "kernel": f"polyhip::k3p::sw_pk_kernel<152,false,{'true' if half else 'false'}> (+ locate + tie wave, all inside score_pass_ms)",
                     "derivation": f"packed {'half-float' if half else 'int16'} recurrence: {per_blk} VALU instructions = "
                                   f"{4 * per_blk:.1f} issue cycles per 512 cells (64 lanes x 2 pairs x 4 columns), 1024 SIMDs x "
                                   f"{clock / 1e9:.3f} GHz (the clock the kernel sustains); HBM is not the bound (166 B per 750,000 cells)",
                     "source": _k3.get("source"),
                     "floor": {"instructions_per_row_block": _k3["floor_instructions_per_row_block"],
                               "ceiling_T_cell_updates_per_s": ceiling_floor / 1e12,
                               "frac": cells / ms_score * 1e3 / ceiling_floor},
                     "hbm_achieved_GBs": alg / ms_score * 1e3 / 1e9},
        "mean_score": float(score.double().mean()), "mean_alignment_len": float(ln.double().mean()),
    }


def tm_scan(dev, n: int = 5_000_000, Lmin: int = 18, Lmax: int = 30):
    """config 5 (one GPU's view: the whole genome): SantaLucia of every Lmin..Lmax-mer"""
    g = torch.empty(n, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC5, g)
    ns, nl = n - Lmin + 1, Lmax - Lmin + 1
    out = [torch.zeros(nl * ns, dtype=torch.float64, device=dev) for _ in range(3)]
    ms = _time(lambda: primers.santalucia_scan_dev(g, n, 0, ns, Lmin, Lmax, 500e-9, 50e-3, 0.0, *out, ns), 10)
    win = sum(n - L + 1 for L in range(Lmin, Lmax + 1))
    gbs = (win * 24 + n) / ms * 1e3 / 1e9
    return {"workload": f"SantaLucia Tm/dH/dS of all {Lmin}..{Lmax}-mers of a {n} B genome (BASELINE configs[4])",
            "windows_per_s": win / ms * 1e3, "ms": ms, "algorithmic_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}


def distance(dev, nfam: int = 1000, copies: int = 100, rows_div: int = 8):
    """config 3, one rank's share: N sketches, this rank's N/rows_div rows against all N columns"""
    s = 1000
    sk = family_sketches(dev, nfam, copies, 10_000, 21, s, 0xC3)
    N = sk.shape[0]
    nrows = N // rows_div
    counts = torch.empty((nrows, N), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(nrows, s, N, s), dtype=torch.uint8, device=dev)
    ms = _time(lambda: mash.shared_counts_dev(sk[:nrows], sk, counts, work), 3)
    # the same row block against an index that is already there (the other row blocks of the matrix, a resident database)
    ms_index = _time(lambda: mash.index_build_dev(sk, work), 3)
    ms_join = _time(lambda: mash.shared_counts_reuse_dev(sk[:nrows], sk, counts, work), 3)
    dist = torch.empty((nrows, N), dtype=torch.float64, device=dev)
    ms_d = _time(lambda: mash.distance_from_counts_dev(counts, s, s, dist), 3)
    mode = mash.shared_counts_mode(work)
    nonzero = int((counts != 0).sum())
    pairs = nrows * N
    out = {"workload": f"{nrows} x {N} sketch pairs (row block 1/{rows_div} of the all-vs-all over {N} sketches of s={s}; "
                       f"{nfam} families x {copies} copies at 1 % substitution) (BASELINE configs[2], one rank)",
           "pairs_per_s_counts": pairs / ms * 1e3, "counts_ms": ms, "index_build_ms": ms_index, "join_only_ms": ms_join,
           "pairs_per_s_join_only": pairs / ms_join * 1e3,
           "pairs_per_s_counts_plus_fp64_distance": pairs / (ms + ms_d) * 1e3, "distance_ms": ms_d,
           "algorithmic_GBs_fp64_out": pairs * 8 / (ms + ms_d) * 1e3 / 1e9,
           # 2 B of u16 count per ordered pair is the algorithmic traffic of the counts call
           "roofline": {"bound": "hbm", "achieved": pairs * 2 / ms * 1e3 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": pairs * 2 / ms * 1e3 / 1e9 / HBM_PEAK_GBS,
                        "frac_join_only": pairs * 2 / ms_join * 1e3 / 1e9 / HBM_PEAK_GBS,
                        "kernel": "polyhip::k2::rowjoin_dense_kernel<10> (+ the index build in counts_ms)"},
           "join_mode": mode[0], "nonzero_pairs": nonzero}
    # the path's collective through the C ABI on a 1-rank communicator (libpolyhip's own RCCL calls; at N ranks
    # bench.py --gpus N times the real exchange) and the index built as 8 parts, as 8 ranks would
    try:
        from . import comm as pcomm
        c = pcomm.Comm(pcomm.unique_id(), 0, 1)
        buf = torch.empty_like(sk[:nrows])
        ms_ag = _time(lambda: c.allgather_sketches(sk[:nrows], buf), 3)
        ms_part = _time(lambda: mash.index_build_part_dev(sk, 3, 8, work), 3)
        c.close()
        out["comm_one_rank"] = {"via": "polyhip_comm_unique_id / _init_rank / polyhip_allgather_sketches_dev (RCCL, C ABI)",
                                "allgather_ms": ms_ag, "bytes": int(buf.numel() * 4), "copy_equal": bool(torch.equal(buf, sk[:nrows])),
                                "index_part_1_of_8_ms": ms_part}
        del buf
    except Exception as e:  # reported, never fatal for the other numbers
        out["comm_one_rank"] = {"error": f"{type(e).__name__}: {e}"}
    del dist, counts, work
    torch.cuda.empty_cache()
    # the whole matrix on one GPU: one index, every row
    counts = torch.empty((N, N), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(N, s, N, s), dtype=torch.uint8, device=dev)
    ms_full = _time(lambda: mash.shared_counts_dev(sk, sk, counts, work), 3, 2)
    out["full_matrix_one_gpu"] = {"pairs": N * N, "ms": ms_full, "pairs_per_s": N * N / ms_full * 1e3,
                                  "u16_GBs": N * N * 2 / ms_full * 1e3 / 1e9,
                                  "self_pairs_share_all_hashes": bool((counts.diagonal() == s).all())}
    return out


def rotation(dev, n: int = 100_000, L: int = 5000):
    seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0x5EED, seqs)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    rot = torch.zeros(n, dtype=torch.int64, device=dev)
    out = torch.zeros_like(seqs)
    ms = _time(lambda: seqhash.least_rotation_batch_dev(seqs, offs, L, rot, out), 5)
    return {"workload": f"RotateSequence of {n} circular sequences of {L} bp", "bases_per_s": n * L / ms * 1e3, "ms": ms,
            "algorithmic_GBs": (2 * n * L + 8 * n) / ms * 1e3 / 1e9}


def hashing(dev, n: int = 100_000, L: int = 5000):
    """seqhash.Hash of n circular double-stranded DNA sequences (normalise, revcomp, 2 rotations, choose, BLAKE3)"""
    seqs = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0x5EED, seqs)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    out = torch.zeros((n, 72), dtype=torch.uint8, device=dev)
    err = torch.zeros(n, dtype=torch.int32, device=dev)
    work = torch.empty(seqhash.seqhash_workspace_bytes(n, n * L, True, True), dtype=torch.uint8, device=dev)
    ms = _time(lambda: seqhash.seqhash_batch_dev(seqs, offs, n * L, L, 0, True, True, out, err, work), 3)
    return {"workload": f"seqhash.Hash(DNA, circular, double-stranded) of {n} sequences of {L} bp",
            "sequences_per_s": n / ms * 1e3, "bases_per_s": n * L / ms * 1e3, "ms": ms}


def fastq_feeder(dev, n: int = 200_000, L: int = 1000):
    """FASTQ image (n records of L bp) -> packed batch, parsed on the device"""
    import numpy as np
    rng = np.random.default_rng(1)
    rec = (b"@r0000000 ch=1 start=2\n" + bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)) + b"\n+\n" +
           bytes(rng.integers(33, 74, L, dtype=np.uint8)) + b"\n")
    img = torch.from_numpy(np.frombuffer(rec * n, np.uint8).copy()).to(dev)
    nb = img.numel()
    seqs = torch.empty(nb, dtype=torch.uint8, device=dev)
    offs = torch.zeros(nb // 7 + 2, dtype=torch.int64, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    work = torch.empty(fastq.workspace_bytes(nb), dtype=torch.uint8, device=dev)
    ms = _time(lambda: fastq.pack_dev(img, seqs, offs, None, res, work), 5)
    r = [int(x) for x in res.cpu()]
    return {"workload": f"FASTQ image of {n} records x {L} bp ({nb} B) -> packed read batch on the device",
            "file_GBs": nb / ms * 1e3 / 1e9, "records_per_s": n / ms * 1e3, "ms": ms, "records": r[0], "error_code": r[1]}


def sw_pairs(dev, n: int = 200_000, L: int = 150):
    """reads against reads: n pairs of L x L (B = A with 5 % substitutions), every pair its own B"""
    a = alphabet.NewAlphabet(list("-ACGT"))
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
    A = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xA2, A)
    B = A.clone().view(n, L)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xA2)
    hit = torch.rand(B.shape, device=dev, generator=gen) < 0.05
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    B[hit] = lut[torch.randint(0, 4, (int(hit.sum()),), device=dev, generator=gen)]
    B = B.reshape(-1).contiguous()
    off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    score = torch.zeros(n, dtype=torch.int64, device=dev)
    ea, eb, er, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4))
    work = torch.empty(align.sw_workspace_bytes(sc, n, L, L, False), dtype=torch.uint8, device=dev)
    ms_score = _time(lambda: align.sw_batch_dev(sc, A, off, L, B, off, L, score, ea, eb, er, work), 3)
    stride = align.sw_traceback_stride(sc, L, L)
    tbw = torch.empty(align.sw_traceback_workspace_bytes(sc, n, L, L), dtype=torch.uint8, device=dev)
    alnA = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    alnB = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    ms_tb = _time(lambda: align.sw_traceback_dev(sc, A, off, L, B, off, L, ea, eb, er, alnA, alnB, ln, tbw, score_t=score), 2)
    cells = n * L * L
    return {"workload": f"{n} pairs of {L} x {L} bp, each with its own B, NUC_4, gap -2",
            "cell_updates_per_s": cells / ms_score * 1e3, "score_pass_ms": ms_score,
            "cell_updates_per_s_with_traceback": cells / (ms_score + ms_tb) * 1e3, "traceback_ms": ms_tb,
            "mean_score": float(score.double().mean())}


def nw(dev, n: int = 200_000, L: int = 150):
    """global alignment of n pairs of L bp reads (B = A with 5 % substitutions), NUC_4, gap -2"""
    a = alphabet.NewAlphabet(list("-ACGT"))
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
    A = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xA1, A)
    B = A.clone().view(n, L)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xA1)
    hit = torch.rand(B.shape, device=dev, generator=gen) < 0.05
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    B[hit] = lut[torch.randint(0, 4, (int(hit.sum()),), device=dev, generator=gen)]
    B = B.reshape(-1).contiguous()
    off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    score = torch.zeros(n, dtype=torch.int64, device=dev)
    err, ln = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2))
    alnA = torch.zeros((n, 2 * L), dtype=torch.uint8, device=dev)
    alnB = torch.zeros((n, 2 * L), dtype=torch.uint8, device=dev)
    work = torch.empty(align.nw_workspace_bytes(n, L, L), dtype=torch.uint8, device=dev)
    ms = _time(lambda: align.nw_align_dev(sc, A, off, L, B, off, L, score, err, alnA, alnB, ln, work), 3)
    return {"workload": f"NeedlemanWunsch (score + aligned strings) of {n} pairs of {L} x {L} bp, NUC_4, gap -2",
            "cell_updates_per_s": n * L * L / ms * 1e3, "ms": ms, "mean_score": float(score.double().mean()),
            "mean_alignment_len": float(ln.double().mean())}


def fasta_feeder(dev, n: int = 100_000, L: int = 4000, width: int = 80):
    """FASTA image (n records of L bp in lines of `width`) -> packed batch, parsed on the device"""
    import numpy as np
    rng = np.random.default_rng(2)
    body = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
    rec = b">seq0000000 some description\n" + b"\n".join(body[i:i + width] for i in range(0, L, width)) + b"\n"
    img = torch.from_numpy(np.frombuffer(rec * n, np.uint8).copy()).to(dev)
    nb = img.numel()
    seqs = torch.empty(nb, dtype=torch.uint8, device=dev)
    offs = torch.zeros(nb // 7 + 2, dtype=torch.int64, device=dev)
    res = torch.zeros(4, dtype=torch.int64, device=dev)
    work = torch.empty(fasta.workspace_bytes(nb), dtype=torch.uint8, device=dev)
    ms = _time(lambda: fasta.pack_dev(img, seqs, offs, None, res, work), 5)
    r = [int(x) for x in res.cpu()]
    return {"workload": f"FASTA image of {n} records x {L} bp in {width}-column lines ({nb} B) -> packed batch on the device",
            "file_GBs": nb / ms * 1e3 / 1e9, "records_per_s": n / ms * 1e3, "ms": ms, "records": r[0], "error_code": r[1]}


def e2e(dev) -> dict:
    """PCIe-inclusive rates of the host-pointer flavours (what a cgo caller with Go-heap buffers gets):
    pageable numpy memory in, pageable numpy memory out, synchronous calls.  Never the headline value."""
    import numpy as np
    out = {}
    # K1: 200k reads x 10 kb (2 GB in, 0.8 GB out)
    n, L, k, s = 200_000, 10_000, 21, 1000
    d = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC2, d)
    host = d.cpu().numpy()
    del d
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    sk = np.zeros((n, s), dtype=np.uint32)
    ms = _wall(lambda: mash.sketch_batch_packed(host, offs, k, s, out=sk), 5, 1)
    out["mash_sketch"] = {"workload": f"polyhip_mash_sketch_batch, {n} reads x {L} B from pageable host memory",
                          "kmers_per_s": n * (L - k) / ms * 1e3, "ms": ms, "pcie_GBs": (n * L + 4 * n * s) / ms * 1e3 / 1e9}
    del host, sk
    # K3: configs[3], host reads in, (score, endA, endB, err) out; then with aligned strings
    a = alphabet.NewAlphabet(list("-ACGT"))
    sc = align.NewScoring(matrix.NewSubstitutionMatrix(a, a, matrix.NUC_4), -2)
    n, LA, LB = 1_000_000, 150, 5000
    B, A = workloads.config4_reads(n, LA, LB, device=dev)
    hA, hB = A.reshape(-1).cpu().numpy(), B.cpu().numpy()
    del A, B
    offA = np.arange(0, (n + 1) * LA, LA, dtype=np.uint64)
    ms = _wall(lambda: align.sw_batch_packed(sc, hA, offA, hB, None), 5, 1)
    out["smith_waterman"] = {"workload": f"polyhip_sw_batch, {n} x {LA} bp reads vs one {LB} bp reference, host pointers",
                             "cell_updates_per_s": n * LA * LB / ms * 1e3, "ms": ms}
    import ctypes as C
    from . import _lib
    L_ = _lib.lib()
    stride = align.sw_traceback_stride(sc, LA, LB)
    o_score, o_len = np.zeros(n, np.int64), np.zeros(n, np.uint32)
    o_ea, o_eb, o_er = (np.zeros(n, np.uint32) for _ in range(3))
    o_alnA, o_alnB = (np.zeros((n, stride), np.uint8) for _ in range(2))

    def sw_strings():
        _lib.check(L_.polyhip_sw_align_batch(sc.handle(), hA.ctypes.data, offA.ctypes.data, n, hB.ctypes.data, None, LB,
                                             o_score.ctypes.data, o_ea.ctypes.data, o_eb.ctypes.data, o_er.ctypes.data,
                                             o_alnA.ctypes.data, o_alnB.ctypes.data, o_len.ctypes.data, stride))
    ms = _wall(sw_strings, 3, 1)
    out["smith_waterman_with_strings"] = {
        "workload": f"polyhip_sw_align_batch, {n} x {LA} bp reads vs one {LB} bp reference, host pointers, aligned strings in "
                    f"{stride}-byte slots (2 x {n * stride / 1e6:.0f} MB cross PCIe into existing pageable buffers)",
        "cell_updates_per_s": n * LA * LB / ms * 1e3, "ms": ms, "mean_alignment_len": float(o_len.mean())}
    del o_alnA, o_alnB
    # the same with PACKED strings (polyhip_sw_align_batch_packed): 0.3 GB instead of 1.05 GB over PCIe
    cap = n * 200
    p_a, p_b, p_off = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(n + 1, np.uint64)

    def sw_packed():
        _lib.check(L_.polyhip_sw_align_batch_packed(sc.handle(), hA.ctypes.data, offA.ctypes.data, n, hB.ctypes.data, None, LB,
                                                    o_score.ctypes.data, o_ea.ctypes.data, o_eb.ctypes.data, o_er.ctypes.data,
                                                    p_a.ctypes.data, p_b.ctypes.data, p_off.ctypes.data, cap))
    ms = _wall(sw_packed, 3, 1)
    out["smith_waterman_with_packed_strings"] = {
        "workload": f"polyhip_sw_align_batch_packed, {n} x {LA} bp reads vs one {LB} bp reference, host pointers, aligned strings "
                    f"compacted on the device ({int(p_off[n]) * 2 / 1e6:.0f} MB cross PCIe)",
        "cell_updates_per_s": n * LA * LB / ms * 1e3, "ms": ms, "string_bytes": int(p_off[n]) * 2}
    del p_a, p_b
    # K4: configs[4], 5 Mb genome in, three fp64 planes (1.56 GB) out into buffers that already exist (a caller that
    # scans more than one genome reuses them; fresh numpy / Go memory adds ~0.1 s of first-touch page faults)
    g = torch.empty(5_000_000, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC5, g)
    hg = g.cpu().numpy()
    del g
    n = len(hg)
    ns = n - 18 + 1
    win = sum(n - L + 1 for L in range(18, 31))
    planes = [np.zeros((13, ns), dtype=np.float64) for _ in range(3)]

    def scan():
        _lib.check(L_.polyhip_santalucia_scan(hg.ctypes.data, n, 18, 30, 500e-9, 50e-3, 0.0, *(p.ctypes.data for p in planes)))
    ms = _wall(scan, 5, 2)
    out["santalucia_scan"] = {"workload": "polyhip_santalucia_scan, 5,000,000 B genome, 18..30-mers, host pointers, "
                                          "three fp64 planes (1.56 GB) into existing pageable buffers",
                              "windows_per_s": win / ms * 1e3, "ms": ms, "pcie_GBs": win * 24 / ms * 1e3 / 1e9}
    ms = _wall(lambda: primers.SantaLuciaScan(hg, 18, 30), 3, 1)
    out["santalucia_scan"]["ms_with_fresh_result_arrays"] = ms
    del planes
    ms = _wall(lambda: primers.SantaLuciaScanFirst(hg, 18, 30, 60.0), 5, 2)
    out["santalucia_scan_first"] = {"workload": "polyhip_santalucia_scan_first, same genome: first length with Tm >= 60 C per start "
                                                "(the pcr grow loop, reduced on the chip: 10 B per start cross PCIe)",
                                    "windows_per_s_equivalent": win / ms * 1e3, "starts_per_s": ns / ms * 1e3, "ms": ms}
    del hg
    # ---- the flavours pipelined in round 3 (chunks through two slots on the calling thread's two streams)
    # K2: 4,000 x 20,000 sketches of s = 1000: the matrix is what crosses PCIe (160 MB of u16 + 640 MB of fp64)
    skd = family_sketches(dev, 200, 100, 10_000, 21, 1000, 0xC3)
    hs_ = skd.cpu().numpy().view(np.uint32)
    del skd
    X = np.ascontiguousarray(hs_[:4000])
    # results go into buffers that already exist (as for the SW strings above: fresh numpy / Go memory adds first-touch page
    # faults at a few GB/s, which is the allocator's time, not the library's)
    o_c, o_d = np.zeros((4000, 20000), np.uint16), np.zeros((4000, 20000), np.float64)

    def dm(with_dist):
        _lib.check(L_.polyhip_mash_distance_matrix(X.ctypes.data, 4000, 1000, hs_.ctypes.data, 20000, 1000, o_c.ctypes.data,
                                                   o_d.ctypes.data if with_dist else None))
    ms_c = _wall(lambda: dm(False), 3, 1)
    ms_cd = _wall(lambda: dm(True), 3, 1)
    out["mash_distance_matrix"] = {"workload": "polyhip_mash_distance_matrix, 4000 x 20000 sketches of s = 1000, host pointers "
                                               "(row blocks of ~64 MB: block b joined while block b-1 crosses PCIe)",
                                   "pairs_per_s_counts": 4000 * 20000 / ms_c * 1e3, "ms_counts": ms_c,
                                   "pairs_per_s_counts_and_fp64": 4000 * 20000 / ms_cd * 1e3, "ms_counts_and_fp64": ms_cd,
                                   "pcie_GBs_counts": (4000 * 20000 * 2 + hs_.nbytes) / ms_c * 1e3 / 1e9,
                                   "pcie_GBs_counts_and_fp64": (4000 * 20000 * 10 + hs_.nbytes) / ms_cd * 1e3 / 1e9}
    del hs_, X, o_c, o_d
    # K5 / S2: 100k circular sequences of 5 kb (500 MB in; rotated: 500 MB out; seqhash: 7.2 MB out)
    nq, Lq = 100_000, 5000
    dq = torch.empty(nq * Lq, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0x5EED, dq)
    hq = dq.cpu().numpy()
    del dq
    oq = np.arange(0, (nq + 1) * Lq, Lq, dtype=np.uint64)
    o_rot, o_seq = np.zeros(nq, np.uint64), np.zeros(nq * Lq, np.uint8)
    ms = _wall(lambda: _lib.check(L_.polyhip_least_rotation_batch(hq.ctypes.data, oq.ctypes.data, nq, o_rot.ctypes.data,
                                                                  o_seq.ctypes.data)), 3, 1)
    out["least_rotation"] = {"workload": f"polyhip_least_rotation_batch, {nq} x {Lq} bp, host pointers, rotated sequences back",
                             "bases_per_s": nq * Lq / ms * 1e3, "ms": ms, "pcie_GBs": 2 * nq * Lq / ms * 1e3 / 1e9}
    del o_seq
    o_h, o_e = np.zeros(nq * 72, np.uint8), np.zeros(nq, np.uint32)
    ms = _wall(lambda: _lib.check(L_.polyhip_seqhash_batch(hq.ctypes.data, oq.ctypes.data, nq, 0, 1, 1, o_h.ctypes.data,
                                                           o_e.ctypes.data)), 3, 1)
    out["seqhash"] = {"workload": f"polyhip_seqhash_batch, {nq} x {Lq} bp (DNA, circular, double-stranded), host pointers",
                      "sequences_per_s": nq / ms * 1e3, "ms": ms, "pcie_GBs": nq * Lq / ms * 1e3 / 1e9}
    # K4 batch: 4M primers of 24 nt (96 MB in + 32 MB of offsets, 96 MB out)
    npz = 4_000_000
    hp = hq[: npz * 24]
    op = np.arange(0, (npz + 1) * 24, 24, dtype=np.uint64)
    o_t = [np.zeros(npz, np.float64) for _ in range(3)]
    ms = _wall(lambda: _lib.check(L_.polyhip_santalucia_batch(hp.ctypes.data, op.ctypes.data, npz, 500e-9, 50e-3, 0.0,
                                                              *(t.ctypes.data for t in o_t))), 3, 1)
    out["santalucia_batch"] = {"workload": f"polyhip_santalucia_batch, {npz} primers of 24 nt, host pointers",
                               "primers_per_s": npz / ms * 1e3, "ms": ms}
    return out


def run(dev) -> dict:
    out = {}
    for name, fn in (("smith_waterman", sw), ("smith_waterman_250bp", lambda d: sw(d, 400_000, 250)),
                     ("smith_waterman_1kb", lambda d: sw(d, 20_000, 1000)), ("smith_waterman_pairs", sw_pairs), ("needleman_wunsch", nw), ("santalucia_scan", tm_scan), ("mash_distance", distance),
                     ("least_rotation", rotation), ("seqhash", hashing), ("fastq_feeder", fastq_feeder),
                     ("fasta_feeder", fasta_feeder), ("e2e_host_pointers", e2e)):
        try:
            out[name] = fn(dev)
        except Exception as e:  # a secondary number must never take the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out
