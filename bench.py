#!/usr/bin/env python3
"""bench.py -- headline benchmark of the poly search hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): k-mers hashed / s through mash.Sketch on
configs[1] -- 1,000,000 synthetic 10 kb reads, k=21, s=1000, per GPU.
A "step" is one pass of K1 (polyhip_mash_sketch_batch_dev) over the whole
read set, inputs already resident in HBM.

Ranks.  `--gpus N` ALWAYS means N ranks, one per GPU.  Started under
torch.distributed.run (WORLD_SIZE set) the script is one of those ranks and
refuses to run if WORLD_SIZE != N.  Started plain (`python bench.py --gpus 8`)
it launches `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...` on itself and returns that job's exit status.
The JSON line carries what actually ran: n_gpus, the torch.distributed
backend, the world size the process group reports, and every rank's device.

Legs in the one JSON line:
  value / roofline   WEAK scaling: every rank sketches its own 1M reads
                (disjoint slice of one splitmix64 stream, no data-path
                collective); timed region = barrier + synchronize on both
                sides, MAX over ranks.
  strong        (N > 1) the SAME 1M reads of configs[1] split over the ranks
                (sharding.shard_range), timed the same way.
  extra         N = 1: secondary rates of every other kernel (SW cell updates/s,
                Tm windows/s, distance pairs/s ...), each with its own CPU baseline
                and, for K1/SW/K4, the PCIe-inclusive host-pointer rate (`e2e`).
                N > 1: configs[2] (per-rank sketches -> one RCCL all-gather ->
                this rank's row block), configs[3] (reads sharded, shared
                reference), configs[4] (window starts sharded with halo).
  cpu_baseline  the CPU oracle's faithful restatement of mash.go:68-104
                (sort-on-accept) on this box: 1 core and all cores, plus the
                tight (insertion) variant and configs[0] (phiX174) wall time.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 10_000
KMER = 21
SKETCH = 1000
SEED = 0xC2
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
WATCHDOG_S = 240      # the N > 1 all-gather leg may not hold the line hostage for longer


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (config: 1,000,000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reads", type=int, default=400, help="reads per core in the CPU-baseline sample")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary kernels")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_extra.json")
                    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.path.join(ROOT, "bench_extra.json"),
                    help="where the FULL record goes (every leg's workload text, roofline, counter sources, CPU samples); "
                         "stdout carries only the compact line (poly_amd/bench_line.py, at most 8 kB)")
    ap.add_argument("--host-devices", type=int, default=0,
                    help="N > 0: the PCIe-inclusive host-pointer legs (extra.e2e_host_pointers) run on the library's device list "
                         "0..N-1 (polyhip_set_devices: ONE host call fanned out over N GPUs; ids wrap around the visible devices)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: become the launcher."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def emit(full: dict, json_fd: int, full_out: str) -> None:
    """The full record -> side file + stderr; the compact line (<= bench_line.LIMIT bytes, json round-trip checked) ->
    the original stdout as the LAST thing written."""
    from poly_amd import bench_line
    try:
        text_full = bench_line.render_full(full)
    except Exception as e:  # the line must still go out
        text_full = json.dumps({"error": f"full record not serialisable: {type(e).__name__}: {e}"})
    try:
        with open(full_out, "w") as f:
            f.write(text_full + "\n")
        full = dict(full, full=os.path.relpath(full_out, ROOT) if full_out.startswith(ROOT) else full_out)
    except OSError as e:
        sys.stderr.write(f"bench.py: could not write {full_out}: {e}\n")
    sys.stderr.write("bench.py full record: " + text_full + "\n")
    sys.stderr.flush()
    try:
        text = bench_line.render(full)
        if len(text) + 1 > bench_line.LIMIT or not isinstance(json.loads(text), dict):
            raise ValueError(f"compact line of {len(text)} bytes")
    except Exception as e:  # whatever happens, a parseable line with the contract keys goes out
        sys.stderr.write(f"bench.py: compact line failed ({type(e).__name__}: {e}); printing the bare contract line\n")
        bare = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                          "scaling", "vs_baseline", "dtype", "data")}
        bare["config"] = {"workload": str((full.get("config") or {}).get("workload"))[:140]}
        rf = full.get("roofline") or {}
        bare["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        text = json.dumps(bare, default=str)
    os.write(json_fd, (text + "\n").encode())


def host_cores() -> int:
    """cores this process may use: the affinity mask, cut down by a cgroup CPU quota if there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(reads_per_core: int):
    """Oracle (port of mash.go:68-104) on host core(s): faithful = full sort on every accepted hash, as
    mash.go:99; tight = one insertion per accepted hash (same result).  1 core and all cores (one thread
    per shard of reads; the reference itself is single-goroutine)."""
    import concurrent.futures as cf

    import numpy as np
    import oracle as orc

    cores = host_cores()

    def run(n_reads: int, faithful: bool, threads: int):
        buf = orc.synth_dna(SEED, n_reads * READ_LEN)
        per = n_reads // threads
        parts = [(buf[t * per * READ_LEN:(t + 1) * per * READ_LEN],
                  np.arange(0, (per + 1) * READ_LEN, READ_LEN, dtype=np.uint64)) for t in range(threads)]
        t0 = time.perf_counter()
        if threads == 1:
            orc.mash_sketch_batch(parts[0][0], parts[0][1], KMER, SKETCH, faithful=faithful)
        else:  # ctypes releases the GIL: these are real threads on the C restatement
            with cf.ThreadPoolExecutor(threads) as ex:
                list(ex.map(lambda p: orc.mash_sketch_batch(p[0], p[1], KMER, SKETCH, faithful=faithful), parts))
        dt = time.perf_counter() - t0
        return per * threads * (READ_LEN - KMER) / dt, dt, per * threads

    v1, dt1, n1 = run(reads_per_core, True, 1)
    out = {
        "value": v1, "unit": "k-mers/s", "cores": 1, "kind": "port",
        "sample": f"first {n1} reads of the same stream ({n1 * (READ_LEN - KMER)} k-mers, {dt1:.1f} s), "
                  "oracle/poly_oracle.c orc_mash_sketch faithful=1 (full sort on accept, as mash.go:99); "
                  "CPU restatement of the reference algorithm (no Go toolchain on this box)",
    }
    vt, dtt, nt = run(reads_per_core * 4, False, 1)
    out["tight_variant"] = {"value": vt, "unit": "k-mers/s", "cores": 1,
                            "sample": f"{nt} reads, insertion instead of sort.Slice ({dtt:.1f} s)"}
    if cores > 1:
        # bounded whatever the core count: 8 / 64 single-core samples' worth of reads, at least one read per thread
        va, dta, na = run(max(cores, reads_per_core * 8 // cores * cores), True, cores)
        out["all_cores"] = {"value": va, "unit": "k-mers/s", "cores": cores,
                            "sample": f"{na} reads, one thread per contiguous shard of reads, faithful ({dta:.1f} s)"}
        vb, dtb, nb = run(max(cores, reads_per_core * 64 // cores * cores), False, cores)
        out["all_cores_tight_variant"] = {"value": vb, "unit": "k-mers/s", "cores": cores,
                                          "sample": f"{nb} reads, insertion variant ({dtb:.1f} s)"}
    # configs[0]: mash.Sketch on phiX174, k=21 s=1000, CPU only (plumbing)
    px = os.path.join(ROOT, "tests", "golden", "phix174.seq")
    if os.path.exists(px):
        seq = open(px, "rb").read().strip()
        reps = 20
        m = orc.Mash(KMER, SKETCH)
        t0 = time.perf_counter()
        for _ in range(reps):
            m.Sketches[:] = 0
            m.Sketch(seq, faithful=True)
        dt = (time.perf_counter() - t0) / reps
        out["config0_phix174"] = {"ms_per_sketch": dt * 1e3, "kmers_per_s": (len(seq) - KMER) / dt, "cores": 1,
                                  "sketch0": int(m.Sketches[0]), "sample": f"{len(seq)} bp, faithful, mean of {reps}"}
    return out


def cpu_baseline_extra():
    """The oracle timed on bounded samples of the secondary workloads (a few seconds each, one host core):
    what the reference's per-call API costs on the CPU for the same shapes as poly_amd/bench_extra.py."""
    import numpy as np
    import oracle as orc
    from poly_amd import workloads
    out = {}

    def entry(units, unit, dt, sample):
        return {"value": units / dt, "unit": unit, "cores": 1, "kind": "port", "sample": f"{sample} ({dt:.1f} s)"}

    # SmithWaterman, configs[3]: 150 bp mutated reads vs one 5 kb reference, NUC_4, gap -2 (align.go:171-232)
    ref, reads = workloads.config4_reads(300)
    ref = bytes(ref)
    rng = np.random.default_rng(0xC4)
    om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
    t = time.perf_counter()
    for r in reads:
        orc.smith_waterman(bytes(r), ref, om, -2)
    out["smith_waterman"] = entry(len(reads) * 150 * 5000, "cell updates/s", time.perf_counter() - t,
                                  f"first {len(reads)} reads of the configs[3] generator vs the 5000 bp reference, "
                                  "orc_smith_waterman (full int64 matrix + traceback, as align.go:171-232)")
    starts = rng.integers(0, 5000 - 150, size=300)
    pairs = [(ref[a:a + 150], bytes(orc.synth_dna(int(a) + 7919 * r, 150))) for r in range(15) for a in starts]
    t = time.perf_counter()
    for x, y in pairs:
        orc.needleman_wunsch(x, y, om, -2)
    out["needleman_wunsch"] = entry(len(pairs) * 150 * 150, "cell updates/s", time.perf_counter() - t,
                                    f"{len(pairs)} pairs of 150 x 150 bp, orc_needleman_wunsch (align.go:100-166)")
    # SantaLucia scan, configs[4] shape on a 400 kb slice
    g = orc.synth_dna(0xC5, 400_000)
    t = time.perf_counter()
    orc.santalucia_scan(g, 18, 30, 500e-9, 50e-3, 0.0)
    nwin = sum(len(g) - L + 1 for L in range(18, 31))
    out["santalucia_scan"] = entry(nwin, "windows/s", time.perf_counter() - t,
                                   "all 18..30-mers of a 400000 B genome, one primers.SantaLucia restatement per window")
    # Distance, configs[2] shape: sorted sketches of s=1000
    sk = np.sort(rng.integers(0, 1 << 29, size=(600, 1000), dtype=np.uint32), axis=1)
    t = time.perf_counter()
    orc.mash_distance_matrix(sk, sk)
    out["mash_distance"] = entry(600 * 600, "pairs/s", time.perf_counter() - t,
                                 "600 x 600 sketches of s=1000, (*Mash).Distance restatement per ordered pair")
    # RotateSequence / Hash of 5 kb circular sequences
    ns = 8000
    blob = bytes(orc.synth_dna(0x5EED, ns * 5000))
    seqs = [blob[i * 5000:(i + 1) * 5000] for i in range(ns)]
    t = time.perf_counter()
    for q in seqs:
        orc.rotate_sequence(q)
    out["least_rotation"] = entry(ns * 5000, "bases/s", time.perf_counter() - t, f"{ns} sequences of 5000 bp, Booth restatement")
    t = time.perf_counter()
    for q in seqs:
        orc.seqhash(q, "DNA", True, True)
    out["seqhash"] = entry(ns, "sequences/s", time.perf_counter() - t,
                           f"{ns} sequences of 5000 bp, seqhash.Hash(DNA, circular, double-stranded) restatement")
    # FASTQ feeder: the restated io/fastq parser (pure Python, so an upper bound on the gap to Go)
    from oracle import fastq_ref
    rec = (b"@r0000000 ch=1 start=2\n" + bytes(rng.choice(list(b"ACGT"), 1000).astype(np.uint8)) + b"\n+\n" +
           bytes(rng.integers(33, 74, 1000, dtype=np.uint8)) + b"\n")
    img = rec * 20000
    t = time.perf_counter()
    fastq_ref.parse_all(img)
    e = entry(len(img) / 1e9, "file GB/s", time.perf_counter() - t,
              "20000 records x 1000 bp through oracle/fastq_ref.py (fastq.go:84-216 restated in Python)")
    e["kind"] = "port (interpreted Python: not comparable with compiled Go)"
    out["fastq_feeder"] = e
    return out


def oracle_spot_checks(extra: dict) -> dict:
    """AFTER the timing: what the secondary legs computed against the CPU oracle (this file is the only place outside tests/
    that imports it; poly_amd/bench_extra.py only hands over the samples).  SW: 8 pairs of the configs[3] batch -- score, end
    cell, both aligned strings (align.go:171-232); K2: 8 x 8 cells of the configs[2] row block incl. far columns
    (mash.go:107-135); K4: 64 windows of the configs[4] genome, Tm / dH / dS bit for bit (primers.go:70-105)."""
    import numpy as np
    import oracle as orc
    res = {}
    sw = extra.get("smith_waterman") if isinstance(extra, dict) else None
    for name in ("smith_waterman", "smith_waterman_250bp", "smith_waterman_500bp", "smith_waterman_1kb"):
        leg = extra.get(name) if isinstance(extra, dict) else None
        spot = leg.pop("_spot", None) if isinstance(leg, dict) else None
        if spot is None:
            continue
        om = orc.SubstitutionMatrix("-ACGT", "-ACGT", orc.NUC_4_SCORES)
        ok = True
        for q in spot["pairs"]:
            sc, a, b, ea, eb = orc.smith_waterman(q["read"], spot["ref"], om, spot["gap"])
            ok &= (sc, ea, eb, a.encode("latin-1"), b.encode("latin-1")) == (q["score"], q["endA"], q["endB"], q["alnA"], q["alnB"])
        res[name] = bool(ok)
    leg = extra.get("mash_distance") if isinstance(extra, dict) else None
    spot = leg.pop("_spot", None) if isinstance(leg, dict) else None
    if spot is not None:
        sk = spot["sketches"]
        res["mash_distance"] = bool(all(orc.mash_shared(sk[i], sk[j]) == c for i, j, c in spot["cells"]))
        res["mash_distance_far_columns_checked"] = sum(1 for i, j, c in spot["cells"] if j >= 12_500)
    leg = extra.get("santalucia_scan") if isinstance(extra, dict) else None
    spot = leg.pop("_spot", None) if isinstance(leg, dict) else None
    if spot is not None:
        res["santalucia_scan"] = bool(all(orc.santalucia(w["window"], 500e-9, 50e-3, 0.0) == (w["tm"], w["dH"], w["dS"]) for w in spot))
    for leg in (extra.values() if isinstance(extra, dict) else ()):  # nothing unserialisable may reach the JSON line
        if isinstance(leg, dict):
            leg.pop("_spot", None)
    return res


def valu_issue_ceiling(kmers_per_s: float) -> dict:
    """K1's VALU-issue ceiling: instructions per k-mer from the SQ_INSTS_VALU counter (profiles/k1_issue.json), times the
    average issue cost of the kernel's own instruction mix -- its disassembly histogram priced with the two measured issue
    classes, 2 cycles for plain add/sub/and/or/xor/lshr, 4 for everything else (scripts/valu_mix.py ->
    profiles/valu_mix.json) -- at the clock the kernel actually sustains (SQ_BUSY_CYCLES / duration of the same counter
    pass: a VALU-dense kernel does not hold the 2.4 GHz peak clock).  None of it is measured by this run; `source` says so."""
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
        cyc = mix["kernels"]["K1 polyhip::k1::sketch_slab_kernel<21>"]["cycles_per_valu_instruction"]
        ki = json.load(open(os.path.join(ROOT, "profiles", "k1_issue.json")))
        ipk, clock = ki["valu_instructions_per_kmer"], ki["clock_GHz"] * 1e9
    except Exception as e:  # the line must not die with a missing profile
        return {"error": f"{type(e).__name__}: {e}"}
    ceiling = 1024 * clock * 64 / (ipk * cyc)
    return {"instructions_per_kmer": ipk, "cycles_per_instruction": cyc, "clock_GHz": clock / 1e9, "simds": 1024,
            "ceiling_kmers_per_s": ceiling, "frac": kmers_per_s / ceiling,
            "ceiling_at_peak_clock_kmers_per_s": 1024 * 2.4e9 * 64 / (ipk * cyc),
            "source": f"{ki.get('source')}; instruction mix: profiles/valu_mix.json (scripts/valu_mix.py: "
                      f"{mix['kernels']['K1 polyhip::k1::sketch_slab_kernel<21>']['valu_full_rate']} full-rate + "
                      f"{mix['kernels']['K1 polyhip::k1::sketch_slab_kernel<21>']['valu_half_rate']} half-rate VALU instructions in the "
                      "kernel's per-read loop); NOT measured by this run"}


def timed_steps(step, steps: int, warmup: int, sync_all, world: int, dev):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize; (elapsed MAX over ranks, mean
    per-launch ms from HIP events recorded on the launch stream)."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    sync_all()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record()  # same stream as the launch (torch's current stream)
        step()
        e1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    kern_ms = sum(ts) / max(1, steps)
    timed_steps.last_median_ms = 0.5 * (ts[(steps - 1) // 2] + ts[steps // 2]) if steps else 0.0
    return elapsed, kern_ms


def allgather_distance(dev, rank: int, world: int, one_gpu_test: bool = False):
    """BASELINE configs[2]: N = 100k sketches over `world` ranks.  Each rank sketches its own families; the sketches are
    all-gathered by THE PRODUCT'S collective -- polyhip_comm_unique_id (rank 0; the 128 bytes travel over the
    torch.distributed rendezvous channel) -> polyhip_comm_init_rank -> polyhip_allgather_sketches_dev, RCCL over xGMI
    behind the C ABI, what the Go host calls -- and every rank computes its row block.  Two ways to get the index:
      replicated   every rank builds the whole index of the gathered set (polyhip_mash_shared_counts_dev)
      sharded      rank r builds part r of the index (polyhip_mash_index_build_part_dev), the parts are exchanged with
                   two ragged all-gathers (polyhip_mash_index_allgather_dev), then polyhip_mash_shared_counts_reuse_dev
    both timed; `ms_per_step` is the faster.  BENCH_ONE_GPU_TEST (all ranks on GPU 0, gloo): RCCL refuses two ranks on
    one device, so the gather falls back to torch.distributed and says so in `allgather.via`."""
    import torch
    import torch.distributed as dist
    from poly_amd import bench_extra, comm as pcomm, mash, sharding

    s, total = SKETCH, 100_000
    lo, hi = sharding.shard_range(total // 100, rank, world)  # families of 100 copies, sharded by family
    fam_max = max(sharding.shard_range(total // 100, r, world)[1] - sharding.shard_range(total // 100, r, world)[0]
                  for r in range(world))
    local = bench_extra.family_sketches(dev, hi - lo, 100, READ_LEN, KMER, s, 0xC3 + 1000 * rank)
    n_local = local.shape[0]
    n_pad = fam_max * 100                       # ncclAllGather's contract: the same count on every rank
    ragged = any(sharding.shard_range(total // 100, r, world) != (r * fam_max, (r + 1) * fam_max) for r in range(world))
    sizes = [100 * (sharding.shard_range(total // 100, r, world)[1] - sharding.shard_range(total // 100, r, world)[0])
             for r in range(world)]
    row0 = sum(sizes[:rank])
    N = sum(sizes)

    c, via = None, None
    if not one_gpu_test:
        uid = [pcomm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        torch.cuda.synchronize()
        dist.barrier()                           # torch's communicator is idle while the product's one is in use
        c = pcomm.Comm(uid[0], rank, world)
        via = "polyhip_allgather_sketches_dev (libpolyhip's own RCCL communicator, C ABI)"
    else:
        via = "torch.distributed all_gather_into_tensor over gloo (one-GPU test: RCCL needs one device per rank)"

    send = local
    if n_local != n_pad:
        send = torch.zeros((n_pad, s), dtype=local.dtype, device=dev)
        send[:n_local] = local
    recv = torch.empty((world * n_pad, s), dtype=local.dtype, device=dev)
    gathered = recv if not ragged else torch.empty((N, s), dtype=local.dtype, device=dev)
    counts = torch.empty((n_local, N), dtype=torch.int16, device=dev)
    work = torch.empty(mash.shared_counts_workspace_bytes(n_local, s, N, s), dtype=torch.uint8, device=dev)

    def gather():
        if c is not None:
            c.allgather_sketches(send, recv)
        else:
            dist.all_gather_into_tensor(recv, send)
        if ragged:                               # trim the padding (device copies on the same stream)
            o = 0
            for r in range(world):
                gathered[o:o + sizes[r]] = recv[r * n_pad:r * n_pad + sizes[r]]
                o += sizes[r]

    def step_replicated():
        gather()
        mash.shared_counts_dev(gathered[row0:row0 + n_local], gathered, counts, work)

    def step_sharded():
        gather()
        mash.index_build_part_dev(gathered, rank, world, work)
        c.index_allgather(N, s, work)
        mash.shared_counts_reuse_dev(gathered[row0:row0 + n_local], gathered, counts, work)

    def sync():
        torch.cuda.synchronize()
        if c is None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, reps=5, warm=2):
        """wall time per step, MAX over ranks; the ranks meet in the step's own collective, the clock starts after a
        barrier on torch's (otherwise idle) communicator"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        return float(dt.item())

    out = {"workload": f"all-vs-all shared counts over {N} sketches (s={s}) sharded by rows over {world} GPUs, "
                       "per-rank sketches all-gathered once per step (BASELINE configs[2])"}
    t_gather = timed(gather, reps=10, warm=3)
    recv_bytes = (world - 1) * n_pad * s * 4
    out["allgather"] = {"via": via, "bytes_sent_per_rank": n_pad * s * 4, "bytes_received_per_rank": recv_bytes,
                        "ms": t_gather * 1e3, "GBs_received_per_rank": recv_bytes / t_gather / 1e9,
                        "bus_GBs": recv_bytes / t_gather / 1e9}
    t_rep = timed(step_replicated)
    ok_rep = bool((counts[:, row0:row0 + n_local].diagonal() == s).all())
    keep = counts.clone()
    out["replicated_index"] = {"ms_per_step": t_rep * 1e3, "pairs_per_s": N * N / t_rep,
                               "self_pairs_share_all_hashes": ok_rep}
    best = t_rep
    if c is not None:
        # every rank must reach the collectives of the sharded step or none: agree first
        flag = torch.ones(1, dtype=torch.int64, device=dev)
        try:
            mash.index_build_part_dev(gathered, rank, world, work)
            torch.cuda.synchronize()
        except Exception as e:  # pragma: no cover - reported, not raised
            flag.zero_()
            out["sharded_index"] = {"error": f"{type(e).__name__}: {e}"}
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        if int(flag.item()) == 1:
            t_sh = timed(step_sharded)
            same = bool(torch.equal(counts, keep))
            it, st = mash.index_part_spans(N, s, world, work)
            out["sharded_index"] = {"ms_per_step": t_sh * 1e3, "pairs_per_s": N * N / t_sh,
                                    "row_block_equals_replicated": same,
                                    "index_bytes_received_per_rank": int((it[-1] - it[0]) - (it[rank + 1] - it[rank])
                                                                         + (st[-1] - st[0]) - (st[rank + 1] - st[rank])),
                                    "via": "polyhip_mash_index_build_part_dev + polyhip_mash_index_allgather_dev "
                                           "(two grouped-ncclBroadcast ragged all-gathers)"}
            if same:
                best = min(best, t_sh)
        elif "sharded_index" not in out:
            out["sharded_index"] = {"error": "another rank failed to build its part"}
        c.close()
    else:
        out["sharded_index"] = {"skipped": "needs the RCCL communicator (one device per rank)"}
    out.update({"pairs_per_s": N * N / best, "ms_per_step": best * 1e3, "allgather_bytes_per_rank": n_pad * s * 4,
                "self_pairs_share_all_hashes": ok_rep})
    return out


def sharded_tm_scan(dev, rank: int, world: int):
    """configs[4]: the 5 Mb genome's window starts split over the ranks (sharding.scan_shard; the 29-byte
    right halo is read from the same replicated genome buffer), no collective."""
    import torch
    import torch.distributed as dist
    from poly_amd import mash, primers, sharding
    n, Lmin, Lmax = 5_000_000, 18, 30
    g = torch.empty(n, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(0xC5, g)
    start0, ns = sharding.scan_shard(n, Lmin, rank, world)
    nl = Lmax - Lmin + 1
    out = [torch.zeros(nl * ns, dtype=torch.float64, device=dev) for _ in range(3)]

    def step():
        primers.santalucia_scan_dev(g, n, start0, ns, Lmin, Lmax, 500e-9, 50e-3, 0.0, *out, ns)
    for _ in range(5):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    win = sum(n - L + 1 for L in range(Lmin, Lmax + 1))
    return {"workload": f"SantaLucia Tm/dH/dS of all {Lmin}..{Lmax}-mers of a {n} B genome, window starts sharded over "
                        f"{world} GPUs (BASELINE configs[4])",
            "windows_per_s": win / float(dt.item()), "ms_per_step": float(dt.item()) * 1e3,
            "this_rank_starts": [int(start0), int(ns)]}


def main() -> int:
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)

    # Anything a library prints on stdout (gloo / RCCL banners, runtime notices) goes to stderr: stdout carries the ONE
    # JSON line, written by rank 0 to the original descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); refusing to report "
                  "a number for a different GPU count", file=sys.stderr)
        return 2
    one_gpu_test = os.environ.get("BENCH_ONE_GPU_TEST") == "1"
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        return 3
    if world > torch.cuda.device_count() and not one_gpu_test:
        if rank == 0:
            print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        return 2
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            # testing aid: all ranks share GPU 0 and talk over gloo (exercises the N > 1 code on a 1-GPU box)
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from poly_amd import mash, sharding

    # who is actually here: every rank's (rank, device index, name, bus id) -- reported, not assumed
    props = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(local_rank),
          "pci_bus_id": getattr(props, "pci_bus_id", None), "hbm_GiB": round(props.total_memory / 2**30, 1)}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        # a collective that only succeeds if all `world` ranks are in the group: sum of 1 over ranks
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        assert int(one.item()) == world == dist.get_world_size(), "process group does not span --gpus ranks"

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n = args.reads
    seqs = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
    mash.synth_dna_dev(SEED, seqs, first=rank * n * READ_LEN)  # rank's slice of one stream
    offs = torch.arange(0, (n + 1) * READ_LEN, READ_LEN, dtype=torch.int64, device=dev)
    out = torch.zeros((n, SKETCH), dtype=torch.int32, device=dev)

    elapsed, kern_ms = timed_steps(lambda: mash.sketch_batch_dev(seqs, offs, KMER, SKETCH, out),
                                   args.steps, args.warmup, sync_all, world, dev)
    kern_ms_median = timed_steps.last_median_ms
    kmers_per_step = n * (READ_LEN - KMER)
    value = world * kmers_per_step * args.steps / elapsed
    alg_bytes = n * (READ_LEN + 4 * SKETCH)  # per launch: reads in, sketches out
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    # parity spot check inside the bench (rank 0 only): 8 reads against the oracle's FAITHFUL variant -- the first read, the
    # LAST one (the slab kernel's persistent workgroups stride to it) and six drawn from the whole batch by a seed that
    # changes with the date and --steps, so that successive runs look at different reads (round-4 verdict: reads 0..7 of
    # one stream, the same eight every round, said nothing about read 999,999).  The reads are taken from the device
    # buffer the kernel read; that the buffer IS the stream's bytes is checked on the first read.
    parity = None
    if rank == 0:
        import datetime
        import numpy as np
        import oracle as orc
        day = datetime.date.today()
        spot_seed = day.toordinal() * 1000 + args.steps
        pick = np.unique(np.concatenate([[0, n - 1], np.random.default_rng(spot_seed).integers(0, n, 6)]))
        host = np.concatenate([seqs[int(r) * READ_LEN:(int(r) + 1) * READ_LEN].cpu().numpy() for r in pick])
        stream_ok = bool((host[:READ_LEN] == orc.synth_dna(SEED, READ_LEN)).all())
        want = orc.mash_sketch_batch(host, np.arange(0, (len(pick) + 1) * READ_LEN, READ_LEN, dtype=np.uint64), KMER, SKETCH,
                                     faithful=True)
        got = out[torch.from_numpy(pick).to(dev)].cpu().numpy().view(np.uint32)
        parity = bool((got == want).all()) and stream_ok
        parity_detail = {"reads": [int(r) for r in pick], "seed": int(spot_seed), "oracle": "orc_mash_sketch faithful=1"}

    # strong scaling: the SAME n reads of configs[1] (stream positions 0 .. n*READ_LEN) split over the ranks
    strong = None
    if world > 1:
        lo, hi = sharding.shard_range(n, rank, world)
        m_loc = hi - lo
        mash.synth_dna_dev(SEED, seqs[:m_loc * READ_LEN], first=lo * READ_LEN)
        s_offs, s_out = offs[:m_loc + 1], out[:m_loc]
        s_elapsed, s_kern = timed_steps(lambda: mash.sketch_batch_dev(seqs[:m_loc * READ_LEN], s_offs, KMER, SKETCH, s_out),
                                        args.steps, args.warmup, sync_all, world, dev)
        strong = {"scaling": "strong", "value": kmers_per_step * args.steps / s_elapsed, "unit": "k-mers/s",
                  "ms_per_step": s_elapsed / args.steps * 1e3, "this_rank_kernel_ms": s_kern,
                  "workload": f"the same {n} reads (configs[1]) split over {world} ranks by sharding.shard_range, "
                              "no data-path collective"}

    traffic, traffic_source = None, None
    tp = os.path.join(ROOT, "profiles", "k1_traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            traffic = tj.get("hbm_bytes_per_launch_1M_reads")
            traffic_source = (f"profiles/k1_traffic.json (round {tj.get('round')}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                              "passes of this command, corrected per MI355X_MICROARCH.md; NOT measured by this run)")
            if traffic is not None and n != 1_000_000:
                traffic = traffic * n / 1_000_000
        except Exception:
            traffic = None

    line = {
        "metric": "mash.Sketch k-mers hashed/s", "value": value, "unit": "k-mers/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"mash.Sketch over {n} synthetic {READ_LEN} B reads per GPU, k={KMER} s={SKETCH} "
                               f"(BASELINE configs[1]), splitmix64 seed 0x{SEED:X}",
                   "reads_per_gpu": n, "read_len": READ_LEN, "k": KMER, "s": SKETCH,
                   "parallelism": f"reads sharded over {world} GPU(s), no data-path collective"},
        "launch": {"ranks": world, "backend": backend or "none (single process)",
                   "group_world_size": dist.get_world_size() if world > 1 else 1,
                   "self_launched": os.environ.get("BENCH_SELF_LAUNCHED") == "1",
                   "one_gpu_test": one_gpu_test, "devices": ranks},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "polyhip::k1::sketch_slab_kernel<21>", "kernel_ms": kern_ms,
                     "kernel_ms_median": kern_ms_median,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     # what actually bounds K1: VALU issue -- priced by valu_issue_ceiling() below
                     "valu_issue": valu_issue_ceiling(kmers_per_step / (kern_ms * 1e-3))},
        "parity_spot_check": parity,
    }
    if rank == 0:
        line["roofline"]["parity_spot_check_reads"] = parity_detail
    if strong is not None:
        line["strong"] = strong

    # free the headline's buffers before the secondary kernels allocate theirs
    del seqs, out, offs
    torch.cuda.empty_cache()
    if world > 1 and not args.no_extra:
        # BASELINE configs[3] sharded: every rank aligns its own 1M reads against the shared reference
        # (pairs split N/G, no collective); the slowest rank sets the time
        r, sw_err = None, None
        try:
            from poly_amd import bench_extra
            r = bench_extra.sw(dev, shard=rank)
            times = [r["score_pass_ms"], r["score_pass_ms"] + r["traceback_ms"]]
        except Exception as e:  # the collective below must still be entered by every rank
            sw_err, times = f"{type(e).__name__}: {e}", [float("inf"), float("inf")]
        t = torch.tensor(times, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if sw_err is None and bool(torch.isfinite(t).all()):
            cells = world * 1_000_000 * 150 * 5000
            sw_extra = {"workload": f"{world} x (" + r["workload"] + "), reads sharded over the GPUs",
                        "cell_updates_per_s": cells / float(t[0].item()) * 1e3, "score_pass_ms": float(t[0].item()),
                        "cell_updates_per_s_with_traceback": cells / float(t[1].item()) * 1e3}
        else:
            sw_extra = {"error": sw_err or "another rank failed"}
        try:
            tm_extra = sharded_tm_scan(dev, rank, world)
        except Exception as e:
            tm_extra = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            line["extra"] = {"smith_waterman": sw_extra, "santalucia_scan": tm_extra}
            if "cell_updates_per_s" in sw_extra:
                # the metric's second half at N ranks: every rank's own 1M config-4 reads, the slowest rank sets the time
                rf1 = (r or {}).get("roofline", {}) if isinstance(r, dict) else {}
                peak = rf1.get("peak")
                sec = {"metric": f"align.SmithWaterman cell updates/s (score pass, BASELINE configs[3] per GPU, {world} GPUs, reads sharded, no collective)",
                       "value": sw_extra["cell_updates_per_s"], "unit": "cell updates/s", "ms_per_step": sw_extra["score_pass_ms"],
                       "value_with_strings": sw_extra.get("cell_updates_per_s_with_traceback"),
                       "roofline": {"bound": rf1.get("bound"), "achieved": sw_extra["cell_updates_per_s"] / 1e12,
                                    "peak": peak * world if peak else None, "unit": "T cell updates/s",
                                    "frac": sw_extra["cell_updates_per_s"] / 1e12 / (peak * world) if peak else None,
                                    "kernel": rf1.get("kernel")}}
                line["roofline"].update({"secondary_metric": "SW cell updates/s", "secondary_value": sec["value"],
                                         "secondary_ms_per_step": sec["ms_per_step"], "secondary_bound": rf1.get("bound"),
                                         "secondary_frac": sec["roofline"]["frac"]})
                line["secondary"] = sec
        # BASELINE configs[2], LAST and under a watchdog: per-rank sketches -> the product's RCCL all-gather -> this
        # rank's row block (the one collective on the path).  It drives a second communicator (libpolyhip's
        # own); if a rank fails inside a step the others would wait in a collective for ever -- the watchdog then
        # prints the line with what has been measured and ends every rank, instead of losing the whole run.
        import threading

        def bail():
            if rank == 0:
                line.setdefault("extra", {})["mash_distance_allgather"] = {
                    "error": f"watchdog: the all-gather leg did not finish within {WATCHDOG_S} s (a rank failed or a "
                             "collective hung); every other number in this line was measured before it"}
                emit(line, json_fd, args.full_out)
            os._exit(0)
        dog = threading.Timer(WATCHDOG_S, bail)
        dog.daemon = True
        dog.start()
        try:
            line_extra = allgather_distance(dev, rank, world, one_gpu_test)
        except Exception as e:
            line_extra = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
        if rank == 0:
            line["extra"]["mash_distance_allgather"] = line_extra
    if rank == 0 and world == 1:
        if not args.no_extra:
            try:
                from poly_amd import bench_extra
                hd = None
                if args.host_devices > 0:
                    hd = [i % torch.cuda.device_count() for i in range(args.host_devices)]
                line["extra"] = bench_extra.run(dev, hd)
            except ImportError:
                pass
            if isinstance(line.get("extra"), dict):
                checks = {"mash_sketch": parity}
                try:
                    checks.update(oracle_spot_checks(line["extra"]))
                except Exception as e:  # reported, never fatal for the measured numbers
                    checks["error"] = f"{type(e).__name__}: {e}"
                    for leg in line["extra"].values():
                        if isinstance(leg, dict):
                            leg.pop("_spot", None)
                line["parity_spot_check"] = checks
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_reads)
            if isinstance(line.get("extra"), dict):
                for name, base in cpu_baseline_extra().items():
                    if isinstance(line["extra"].get(name), dict):
                        line["extra"][name]["cpu_baseline"] = base
        # BASELINE's metric has two halves: k-mers/s (value) AND SW cell updates/s.  The second half goes where a reader
        # of the line's head (roofline.*: flat scalars) and of its TAIL (the last object of the line) both find it.
        swx = line.get("extra", {}).get("smith_waterman") if isinstance(line.get("extra"), dict) else None
        if isinstance(swx, dict) and "cell_updates_per_s" in swx:
            rf = swx.get("roofline", {})
            sec = {"metric": "align.SmithWaterman cell updates/s (score pass, BASELINE configs[3]: 1M x 150 bp vs 5 kb, one GPU)",
                   "value": swx["cell_updates_per_s"], "unit": "cell updates/s", "ms_per_step": swx["score_pass_ms"],
                   "value_with_strings": swx.get("cell_updates_per_s_align_one_call"), "ms_per_step_with_strings": swx.get("align_one_call_ms"),
                   "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel")},
                   "cpu_baseline": swx.get("cpu_baseline")}
            line["roofline"].update({"secondary_metric": "SW cell updates/s", "secondary_value": sec["value"],
                                     "secondary_ms_per_step": sec["ms_per_step"], "secondary_bound": rf.get("bound"),
                                     "secondary_frac": rf.get("frac"), "secondary_peak_T_cell_updates_per_s": rf.get("peak")})
            if isinstance(line.get("cpu_baseline"), dict) and isinstance(sec["cpu_baseline"], dict):
                line["cpu_baseline"]["secondary_value"] = sec["cpu_baseline"].get("value")
                line["cpu_baseline"]["secondary_unit"] = "SW cell updates/s, 1 core"
            line["secondary"] = sec
    if rank == 0:
        # the line ends with what a reader of its last two kilobytes needs: both halves of the metric and the checks
        for key in ("secondary", "parity_spot_check"):
            if key in line:
                line[key] = line.pop(key)
        try:
            from poly_amd.bench_extra import traffic_checks
            tc = traffic_checks(line)
        except Exception as e:
            tc = {"error": f"{type(e).__name__}: {e}"}
        if tc.get("failed"):
            sys.stderr.write("bench.py: counter traffic below the algorithmic bytes: " + "; ".join(tc["failed"]) + "\n")
        line["summary"] = {"kmers_per_s": line["value"], "hbm_frac": line["roofline"]["frac"], "traffic_checks": tc,
                           "sw_cell_updates_per_s": (line.get("secondary") or {}).get("value"),
                           "sw_valu_frac": ((line.get("secondary") or {}).get("roofline") or {}).get("frac"),
                           "parity_spot_check": line.get("parity_spot_check")}
        sys.stdout.flush()
        emit(line, json_fd, args.full_out)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
