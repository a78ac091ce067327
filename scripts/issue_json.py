#!/usr/bin/env python3
"""profiles/k1_issue.json, k3_issue.json and k1_traffic.json from the round's rocprofv3 summaries (rocpd_summary.py tables), so
that bench.py's `roofline.valu_issue` / `traffic` fields carry THIS round's counters and nobody copies numbers by hand:

    python scripts/issue_json.py r05 profiles

k1_issue   <- {rnd}_k1_issue.md   (scripts/quick_k1.py under --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES ...: launches of 100,000 reads)
k3_issue   <- {rnd}_k3_pmc_nooverlap.md (POLYHIP_SW_OVERLAP=0 scripts/quick_k3tb.py under --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES ...)
k1_traffic <- {rnd}_bench_fetch.md + {rnd}_bench_write.md (bench.py --no-extra under --pmc FETCH_SIZE / WRITE_SIZE)"""
import json
import re
import sys


def tables(path):
    stats, pmc = {}, {}
    for line in open(path):
        m = re.match(r"\| `(.*)` \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \| ([0-9.]+) \|", line)
        if m:
            stats[m.group(1)] = {"calls": int(m.group(2)), "total_ms": float(m.group(3)), "mean_ms": float(m.group(4))}
        m = re.match(r"\| `(.*)` \| (\w+) \| (\d+) \| ([0-9.e+]+) \| ([0-9.e+]+) \|", line)
        if m:
            pmc.setdefault(m.group(1), {})[m.group(2)] = {"dispatches": int(m.group(3)), "sum": float(m.group(4)), "per": float(m.group(5))}
    return stats, pmc


def pick(d, sub):
    ks = [k for k in d if sub in k]
    assert len(ks) == 1, (sub, ks)
    return ks[0], d[ks[0]]


def main():
    rnd, d = sys.argv[1], sys.argv[2]
    # ---- K1
    st, pm = tables(f"{d}/{rnd}_k1_issue.md")
    name, c = pick(pm, "sketch_slab_kernel")
    launches = st[name]["calls"]
    kmers = launches * 100_000 * (10_000 - 21)
    k1 = {"round": rnd, "valu_instructions_per_kmer": c["SQ_INSTS_VALU"]["sum"] * 64 / kmers,
          "salu_instructions_per_kmer": c["SQ_INSTS_SALU"]["sum"] * 64 / kmers if "SQ_INSTS_SALU" in c else None,
          "clock_GHz": c["SQ_BUSY_CYCLES"]["per"] / (st[name]["mean_ms"] * 1e-3) / 1e9,
          "source": f"profiles/{rnd}_k1_issue.md (scripts/quick_k1.py under rocprofv3 --pmc, {launches} launches of 100,000 reads): SQ_INSTS_VALU "
                    f"{c['SQ_INSTS_VALU']['sum']:.4e} x 64 lanes / ({launches} x 100,000 reads x 9,979 k-mers); SQ_BUSY_CYCLES "
                    f"{c['SQ_BUSY_CYCLES']['per']:.4e} per shader engine and launch / {st[name]['mean_ms']:.4f} ms; scripts/issue_json.py"}
    json.dump(k1, open(f"{d}/k1_issue.json", "w"), indent=1)
    # ---- K3
    st, pm = tables(f"{d}/{rnd}_k3_pmc_nooverlap.md")
    if any("sw_pk1x2_kernel<76, false>" in k for k in pm):  # round 6: a lane's 152 rows over two lanes, the lower lane one block behind
        name, c = pick(pm, "sw_pk1x2_kernel<76, false>")
        waves, rows, steps = 1_000_000 / 64.0, 76, 1251
    else:
        name, c = pick(pm, "sw_pk1_kernel<152, false>")
        waves, rows, steps = 1_000_000 / 128.0, 152, 1250  # two pairs per lane, 64 lanes
    launches = st[name]["calls"]
    per_launch = c["SQ_INSTS_VALU"]["sum"] / launches
    k3 = {"round": rnd, "kernel": name, "valu_instructions_per_row_block": per_launch / (rows * steps * waves),
          "clock_GHz": c["SQ_BUSY_CYCLES"]["per"] / (st[name]["mean_ms"] * 1e-3) / 1e9, "floor_instructions_per_row_block": 14,
          "source": f"profiles/{rnd}_k3_pmc_nooverlap.md (POLYHIP_SW_OVERLAP=0, 1M pairs, {name}, {launches} launches): SQ_INSTS_VALU "
                    f"{per_launch:.4e} per launch / ({rows} rows x {steps} steps x {waves:.1f} waves); SQ_BUSY_CYCLES {c['SQ_BUSY_CYCLES']['per']:.4e} per "
                    f"shader engine and launch / {st[name]['mean_ms']:.3f} ms; floor = 3 packed ops per cell pair x 4 columns + 2 for the block "
                    "maximum; scripts/issue_json.py"}
    json.dump(k3, open(f"{d}/k3_issue.json", "w"), indent=1)
    # ---- K1 traffic
    _, pf = tables(f"{d}/{rnd}_bench_fetch.md")
    _, pw = tables(f"{d}/{rnd}_bench_write.md")
    name, cf = pick(pf, "sketch_slab_kernel")
    _, cw = pick(pw, "sketch_slab_kernel")
    fkb = cf["FETCH_SIZE"]["sum"] / (cf["FETCH_SIZE"]["dispatches"])
    wkb = cw["WRITE_SIZE"]["sum"] / (cw["WRITE_SIZE"]["dispatches"])
    rd, wr, alg = 2 * fkb * 1024, wkb * 1024, 1_000_000 * (10_000 + 4 * 1000)
    k1t = {"round": rnd, "kernel": f"polyhip::k1::sketch_slab_kernel<21> (the dominant kernel)",
           "workload": "1,000,000 reads x 10,000 B, k=21, s=1000 (bench.py default)", "FETCH_SIZE_KB_per_launch": fkb,
           "WRITE_SIZE_KB_per_launch": wkb,
           "corrections": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section; calibrated in profiles/r01_calib_fetch.md on this access "
                          "width); WRITE_SIZE x1 (profiles/r01_calib_write.md)",
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch_1M_reads": rd + wr,
           "algorithmic_bytes_per_launch": float(alg),
           "sources": [f"profiles/{rnd}_bench_fetch.md", f"profiles/{rnd}_bench_write.md", "profiles/r01_calib_fetch.md", "profiles/r01_calib_write.md",
                       "scripts/issue_json.py"],
           "ratio_to_algorithmic": (rd + wr) / alg}
    json.dump(k1t, open(f"{d}/k1_traffic.json", "w"), indent=1)
    print(json.dumps({"k1_issue": k1, "k3_issue": k3, "k1_traffic_ratio": k1t["ratio_to_algorithmic"]}, indent=1))


if __name__ == "__main__":
    main()
