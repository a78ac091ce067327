"""Secondary rates reported next to bench.py's headline k-mers/s (BASELINE configs 3-5
and seqhash): SW cell updates/s, Tm windows/s, distance pairs/s, least-rotation bases/s.
Device-resident timing with events on the launch stream; synthetic inputs per SURVEY 8d."""
from __future__ import annotations

import torch

from . import align, alphabet, fasta, fastq, mash, matrix, primers, seqhash, workloads

HBM_PEAK_GBS = 8000.0


def _time(fn, reps: int = 10, warm: int = 5) -> float:
    """ms per launch, events on the launch stream, after `warm` (>= 5) untimed launches.  A launch of a
    millisecond or more: median of `reps` (>= 10) separately timed launches.  Below that an event pair around ONE launch
    measures the event machinery as much as the kernel (round 3: 0.33 / 0.36 / 0.46 ms for the same K4 kernel on three
    boxes), so 20 back-to-back launches sit inside one event pair and the figure is the median of >= 10 such groups / 20."""
    reps, warm = max(reps, 10), max(warm, 5)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p0.record()
    fn()
    p1.record()
    torch.cuda.synchronize()
    inner = 20 if p0.elapsed_time(p1) < 1.0 else 1
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) / inner for e0, e1 in evs)
    _time.last_inner = inner
    return 0.5 * (ts[(reps - 1) // 2] + ts[reps // 2])


def _traffic(name: str):
    """HBM bytes per launch of the named leg from the round's PMC passes (profiles/traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs of scripts/collect_profiles_r04.sh, corrected per MI355X_MICROARCH.md) -- counters cannot
    be read from inside this process, so the number is the profile's, not this run's; None when there is no entry."""
    import json
    import os
    try:
        t = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")))
        return t.get(name, {}).get("hbm_bytes_per_launch"), t.get(name, {}).get("source")
    except Exception:
        return None, None


def _hbm_roofline(name: str, alg_bytes: float, ms: float, kernel: str, bound_note: str | None = None) -> dict:
    gbs = alg_bytes / ms * 1e3 / 1e9
    traffic, src = _traffic(name)
    r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "algorithmic_bytes_per_launch": int(alg_bytes), "traffic": traffic, "traffic_source": src, "kernel": kernel,
         "launches_per_timed_group": getattr(_time, "last_inner", 1)}
    r["traffic_ratio"] = traffic / alg_bytes if traffic and alg_bytes else None  # counter bytes / algorithmic bytes, >= ~1
    if bound_note:
        r["bound_note"] = bound_note
    return r


def traffic_checks(node, path="line", out=None) -> dict:
    """Every roofline object of the line that carries both `traffic` (HBM bytes from the PMC passes) and
    `algorithmic_bytes_per_launch`: traffic below 0.98 x the algorithmic bytes is impossible -- a kernel cannot move fewer
    bytes than the bytes it must read and write -- and means the counter selection dropped kernels (round 4: the feeders'
    two templated passes).  Returns {"checked": n, "failed": [paths...]}; bench.py puts it in the line's summary."""
    if out is None:
        out = {"checked": 0, "failed": []}
    if isinstance(node, dict):
        t, a = node.get("traffic"), node.get("algorithmic_bytes_per_launch")
        if isinstance(t, (int, float)) and isinstance(a, (int, float)) and a > 0:
            out["checked"] += 1
            node.setdefault("traffic_ratio", t / a)
            if t < 0.98 * a:
                out["failed"].append(f"{path}: traffic {t:.4g} B < 0.98 x algorithmic {a:.4g} B")
        for k, v in node.items():
            traffic_checks(v, f"{path}.{k}", out)
    return out


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
    # profiles/valu_mix.json) at the clock the kernel sustains; beside it the ceiling of the 14-instruction recurrence
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
    # what bench.py compares with the oracle after the timing (it owns the oracle import; this package never touches it):
    # 8 pairs spread over the batch -- read, score, end cell, both aligned strings of the LAST call above (score + strings
    # in one call)
    idx = [int(i) for i in torch.linspace(0, n - 1, 8).long().tolist()]
    lens = ln[idx].tolist()
    spot = {"ref": bytes(B.cpu().numpy()), "gap": -2, "pairs": [
        {"pair": p, "read": bytes(A[p * LA:(p + 1) * LA].cpu().numpy()), "score": int(score[p]), "endA": int(ea[p]), "endB": int(eb[p]),
         "alnA": bytes(alnA[p, stride - l:].cpu().numpy()), "alnB": bytes(alnB[p, stride - l:].cpu().numpy())}
        for p, l in zip(idx, lens)]}
    if (n, LA, LB) != (1_000_000, 150, 5000):  # other read lengths: the plain figures (which kernels ran is the library's choice)
        return {
            "workload": f"{n} x {LA} bp reads vs one {LB} bp reference, NUC_4, gap -2",
            "cell_updates_per_s": cells / ms_score * 1e3, "score_pass_ms": ms_score,
            "cell_updates_per_s_with_traceback": cells / (ms_score + ms_tb) * 1e3, "traceback_ms": ms_tb,
            "align_one_call_ms": ms_fused, "cell_updates_per_s_align_one_call": cells / ms_fused * 1e3,
            "score_path": align.last_path(), "traceback_path": align.sw_traceback_last_path(),
            "mean_score": float(score.double().mean()), "mean_alignment_len": float(ln.double().mean()), "_spot": spot,
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
                     "kernel": (f"{_k3.get('kernel', 'polyhip::k3p::sw_pk1x2_kernel<76>')} (half-float cells, two lanes per 152 rows) + "
                                "k3p::sw_locate16_kernel<76> + k3w::sw_wave_kernel<3> for the ties, all inside score_pass_ms" if half else
                                "polyhip::k3p::sw_pk_kernel<152,false,false> (int16 cells) + locate + tie wave, all inside score_pass_ms"),
                     "derivation": f"packed {'half-float' if half else 'int16'} recurrence: {per_blk} VALU instructions = "
                                   f"{4 * per_blk:.1f} issue cycles per 512 cells (64 lanes x 2 pairs x 4 columns), 1024 SIMDs x "
                                   f"{clock / 1e9:.3f} GHz (the clock the kernel sustains); HBM is not the bound (166 B per 750,000 cells)",
                     "source": _k3.get("source"),
                     "floor": {"instructions_per_row_block": _k3["floor_instructions_per_row_block"],
                               "ceiling_T_cell_updates_per_s": ceiling_floor / 1e12,
                               "frac": cells / ms_score * 1e3 / ceiling_floor},
                     "hbm_achieved_GBs": alg / ms_score * 1e3 / 1e9},
        "mean_score": float(score.double().mean()), "mean_alignment_len": float(ln.double().mean()), "_spot": spot,
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
    # 64 windows for bench.py's oracle comparison: (start, length) spread over the genome and the lengths
    gen = torch.Generator().manual_seed(5)
    starts = torch.randint(0, n - Lmax, (64,), generator=gen).tolist()
    spot = []
    for q, a in enumerate(starts):
        L = Lmin + q % nl
        spot.append({"start": a, "L": L, "window": bytes(g[a:a + L].cpu().numpy()),
                     "tm": float(out[0][(L - Lmin) * ns + a]), "dH": float(out[1][(L - Lmin) * ns + a]),
                     "dS": float(out[2][(L - Lmin) * ns + a])})
    return {"workload": f"SantaLucia Tm/dH/dS of all {Lmin}..{Lmax}-mers of a {n} B genome (BASELINE configs[4])",
            "windows_per_s": win / ms * 1e3, "ms": ms, "algorithmic_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
            # 24 B out per window + the genome once; this one IS bound by the HBM write stream
            "roofline": _hbm_roofline("santalucia_scan", win * 24 + n, ms, "polyhip::k4::scan_kernel<18,30>",
                                      "HBM write stream: ~6.3 TB/s is what a pure streaming kernel reaches on this part (MI355X_MICROARCH.md)"),
            "_spot": spot}


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
    index_info = mash.index_build_info(work)  # which index build ran (1 = the sliced build) and its geometry
    nonzero = int((counts != 0).sum())
    pairs = nrows * N
    # 8 x 8 cells for bench.py's oracle comparison: 8 rows of the block, each against 4 columns of its own family (hundreds
    # of shared hashes) and 4 FAR columns (other ranks' sketches, beyond the block's own rows)
    gen = torch.Generator().manual_seed(3)
    rows_ = torch.randint(0, nrows, (8,), generator=gen).tolist()
    cells, need = [], set()
    for i in rows_:
        cols = [(i // copies) * copies + int(c) for c in torch.randint(0, copies, (4,), generator=gen)]
        cols += [int(c) for c in torch.randint(nrows, N, (4,), generator=gen)]
        for j in cols:
            cells.append((i, j, int(counts[i, j]) & 0xFFFF))
            need.update((i, j))
    spot = {"s": s, "cells": cells, "sketches": {q: sk[q].cpu().numpy().view("uint32").copy() for q in sorted(need)}}
    out = {"workload": f"{nrows} x {N} sketch pairs (row block 1/{rows_div} of the all-vs-all over {N} sketches of s={s}; "
                       f"{nfam} families x {copies} copies at 1 % substitution) (BASELINE configs[2], one rank)",
           "pairs_per_s_counts": pairs / ms * 1e3, "counts_ms": ms, "index_build_ms": ms_index, "join_only_ms": ms_join,
           "pairs_per_s_join_only": pairs / ms_join * 1e3,
           "pairs_per_s_counts_plus_fp64_distance": pairs / (ms + ms_d) * 1e3, "distance_ms": ms_d,
           "algorithmic_GBs_fp64_out": pairs * 8 / (ms + ms_d) * 1e3 / 1e9,
           # 2 B of u16 count per ordered pair is the algorithmic traffic of the counts call
           "roofline": dict(_hbm_roofline("mash_distance", pairs * 2, ms,
                                          "polyhip::k2::rowjoin_dense_kernel<10, true, true> + the index build in counts_ms "
                                          "(check4 / scatter4 / fine4: the sliced build on 4-byte intermediate items)"),
                            frac_join_only=pairs * 2 / ms_join * 1e3 / 1e9 / HBM_PEAK_GBS),
           "index_build": index_info,
           "join_mode": mode[0], "nonzero_pairs": nonzero, "_spot": spot}
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
            "algorithmic_GBs": (2 * n * L + 8 * n) / ms * 1e3 / 1e9,
            # sequence in, rotated sequence + index out
            "roofline": _hbm_roofline("least_rotation", 2 * n * L + 8 * n, ms, "polyhip::k5::least_rotation_wave_kernel",
                                      "the candidate search compares bytes in LDS: issue-bound well below the HBM line (DESIGN.md K5)")}


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
            "sequences_per_s": n / ms * 1e3, "bases_per_s": n * L / ms * 1e3, "ms": ms,
            # algorithmic: the sequence in, 72 + 4 bytes out; everything between (normalised copy, reverse complement, two
            # rotations) is the implementation's own traffic
            "roofline": _hbm_roofline("seqhash", n * L + 76 * n, ms, "polyhip::s2::prepare / k5 wave x2 / select / chunk / hash",
                                      "BLAKE3 compression is ALU work: see DESIGN.md S2 for the instruction-issue bound")}


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
            "file_GBs": nb / ms * 1e3 / 1e9, "records_per_s": n / ms * 1e3, "ms": ms, "records": r[0], "error_code": r[1],
            # the file in, the sequences + one offset per record out
            "roofline": _hbm_roofline("fastq_feeder", nb + r[3] + 8 * (r[0] + 1), ms, "polyhip::fq::* (count, ranked write, validate, scan, gather)")}


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
            "file_GBs": nb / ms * 1e3 / 1e9, "records_per_s": n / ms * 1e3, "ms": ms, "records": r[0], "error_code": r[1],
            "roofline": _hbm_roofline("fasta_feeder", nb + r[2] + 8 * (r[0] + 1), ms, "polyhip::fq::* (count, ranked write, classify, scans, gather)")}


def e2e(dev, host_devices=None) -> dict:
    """PCIe-inclusive rates of the host-pointer flavours (what a cgo caller with Go-heap buffers gets):
    pageable numpy memory in, pageable numpy memory out, synchronous calls.  Never the headline value.
    `host_devices`: the library's device list for these calls (polyhip_set_devices; bench.py --host-devices N) -- one
    host call fanned out over N GPUs, each shard over its own PCIe link; None = the one-device call."""
    import numpy as np
    from . import devices as _devices
    out = {}
    if host_devices:
        _devices.set_devices(host_devices)
    try:
        out = _e2e_body(dev)
    finally:
        if host_devices:
            _devices.set_devices([])
    for v in out.values():
        if isinstance(v, dict):
            v["devices"] = list(host_devices) if host_devices else "calling thread's current device (no list)"
    return out


def _e2e_body(dev) -> dict:
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


def e2e_fanout_check(dev) -> dict:
    """The fan-out's own cost on this box: the K1 host call of e2e() once on the plain one-device path and once on the
    device list [0, 0] (two workers sharing the GPU and its one PCIe link -- nothing to gain here, so the difference is
    what splitting, the worker hand-off and the second pipeline cost).  On a multi-GPU node: bench.py --host-devices N."""
    import numpy as np
    from . import devices as _devices
    n, L, k, s = 100_000, 10_000, 21, 1000
    d = torch.empty(n * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC2, d)
    host = d.cpu().numpy()
    del d
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    sk1, sk2 = np.zeros((n, s), dtype=np.uint32), np.zeros((n, s), dtype=np.uint32)
    ms1 = _wall(lambda: mash.sketch_batch_packed(host, offs, k, s, out=sk1), 5, 1)
    with _devices.devices([dev.index or 0] * 2):
        ms2 = _wall(lambda: mash.sketch_batch_packed(host, offs, k, s, out=sk2), 5, 1)
    return {"workload": f"polyhip_mash_sketch_batch, {n} reads x {L} B from pageable host memory",
            "ms_one_device": ms1, "ms_device_list_0_0": ms2, "same_sketches": bool((sk1 == sk2).all())}


def e2e_exchange_check(dev) -> dict:
    """BASELINE configs[2] in ONE host call (polyhip_mash_sketch_distance_matrix: reads -> sketches -> index -> matrix rows) on
    this box: the one-device call, the device list [0, 0] with the devices building ONE index together (index items
    exchanged by value range; the sketches are not gathered) and the same list gathering the sketches
    (POLYHIP_K2_EXCHANGE=0).  Two workers share one GPU here, so the exchange can only cost; the figures say what its
    five host round trips and the split kernels cost, and that all three give the same matrix."""
    import os
    import numpy as np
    from . import devices as _devices
    n, L, k, s = 16_000, 10_000, 21, 1000
    d = torch.empty(n // 2 * L, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xE5, d)
    half = d.cpu().numpy().reshape(n // 2, L)
    del d
    rng = np.random.default_rng(5)
    twin = half.copy()
    hit = rng.random(twin.shape) < 0.004
    twin[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    host = np.concatenate([half, twin]).reshape(-1)
    del half, twin, hit
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    res, out = {}, {"workload": f"polyhip_mash_sketch_distance_matrix, {n} reads x {L} B from pageable host memory, k={k}, s={s}, "
                                f"counts only ({n * n * 2 / 1e6:.0f} MB back)"}

    def call(tag):
        t = []
        for _ in range(3):
            import time
            t0 = time.perf_counter()
            _, c, _ = mash.sketch_distance_matrix_packed(host, offs, k, s, want_sketches=False, want_dist=False)
            t.append((time.perf_counter() - t0) * 1e3)
        res[tag] = c
        out["ms_" + tag] = sorted(t)[1]
        out["path_" + tag] = mash.sketch_distance_matrix_last_path()

    call("one_device")
    with _devices.devices([dev.index or 0] * 2):
        call("list_0_0_item_exchange")
        os.environ["POLYHIP_K2_EXCHANGE"] = "0"
        try:
            call("list_0_0_sketch_gather")
        finally:
            os.environ.pop("POLYHIP_K2_EXCHANGE", None)
    out["same_matrix"] = bool((res["one_device"] == res["list_0_0_item_exchange"]).all() and
                              (res["one_device"] == res["list_0_0_sketch_gather"]).all())
    out["nonzero_pairs"] = int((res["one_device"] != 0).sum())
    return out


def run(dev, host_devices=None) -> dict:
    # (the 1 kb leg: 80k pairs fill the chip at 8 lanes per pair -- 6.1e12 cell updates/s; 20k pairs leave it a third full: 4.3e12)
    out = {}
    for name, fn in (("smith_waterman", sw), ("smith_waterman_250bp", lambda d: sw(d, 400_000, 250)),
                     ("smith_waterman_500bp", lambda d: sw(d, 160_000, 500)),
                     ("smith_waterman_1kb", lambda d: sw(d, 80_000, 1000)),
                     ("smith_waterman_pairs", sw_pairs), ("needleman_wunsch", nw), ("santalucia_scan", tm_scan), ("mash_distance", distance),
                     ("least_rotation", rotation), ("seqhash", hashing), ("fastq_feeder", fastq_feeder),
                     ("fasta_feeder", fasta_feeder), ("e2e_host_pointers", lambda d: e2e(d, host_devices)),
                     ("e2e_fanout_check", e2e_fanout_check), ("e2e_exchange_check", e2e_exchange_check)):
        try:
            out[name] = fn(dev)
        except Exception as e:  # a secondary number must never take the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out
