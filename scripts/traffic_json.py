#!/usr/bin/env python3
"""profiles/traffic.json from the round's FETCH_SIZE / WRITE_SIZE passes of scripts/quick_legs.py (rocpd_summary tables):

    python scripts/traffic_json.py r04 profiles > profiles/traffic.json

Per leg: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 summed over the leg's kernels / CALLS -- the
gfx950 corrections of MI355X_MICROARCH.md's HBM section (FETCH_SIZE counts half of what a streaming read moves on this
part: calibrated in profiles/r01_calib_fetch.md; WRITE_SIZE x 1: profiles/r01_calib_write.md)."""
import json
import os
import re
import sys

CALLS = 4
LEGS = {  # leg -> (group, namespace markers: SUBSTRINGS of the kernel name -- round 4 matched on the prefix `polyhip::fq::`,
    # which rocpd_summary.py had cut off the two long templated feeder kernels (`::fq::count_newlines_kernel<...>`), so the
    # two passes that read the whole image were not counted and the feeders' traffic read BELOW their algorithmic bytes)
    "santalucia_scan": ("A", ("::k4::",)),
    "least_rotation": ("A", ("::k5::",)),
    "fastq_feeder": ("A", ("::fq::",)),
    "seqhash": ("B", ("::s2::", "::k5::")),
    "fasta_feeder": ("B", ("::fq::",)),
}


def counter_per_dispatch(path, counter):
    """kernel -> counter value per DISPATCH (the table's last column)"""
    out = {}
    for line in open(path):
        m = re.match(r"\| `(.*)` \| (\w+) \| (\d+) \| ([0-9.e+]+) \| ([0-9.e+]+) \|", line)
        if m and m.group(2) == counter:
            out[m.group(1)] = float(m.group(5))
    return out


def k2_leg(rnd, d):
    """K2's one-shot row block (scripts/quick_k2c.py: one index build, five joins against it, three one-shot calls): every
    `::k2::` kernel runs once per one-shot call, so the call's traffic is the sum of the kernels' per-dispatch values"""
    try:
        f = counter_per_dispatch(f"{d}/{rnd}_k2_fetch.md", "FETCH_SIZE")
        w = counter_per_dispatch(f"{d}/{rnd}_k2_write.md", "WRITE_SIZE")
    except OSError:
        return None
    ks = sorted(set(k for k in list(f) + list(w) if "::k2::" in k))
    rd = 2 * sum(f.get(k, 0.0) for k in ks) * 1024
    wr = sum(w.get(k, 0.0) for k in ks) * 1024
    return {"hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "kernels": ks,
            "per_kernel_MB": {k: round((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024 / 1e6, 1) for k in ks},
            "source": f"profiles/{rnd}_k2_fetch.md + profiles/{rnd}_k2_write.md (rocprofv3 --pmc passes of scripts/quick_k2c.py; per-dispatch "
                      "FETCH_SIZE x 2 and WRITE_SIZE x 1 KB of every polyhip::k2 kernel, summed: one one-shot row block = index build + "
                      "join); NOT measured by the bench run"}


def counter_sums(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"\| `(.*)` \| (\w+) \| (\d+) \| ([0-9.e+]+) \| ([0-9.e+]+) \|", line)
        if m and m.group(2) == counter:
            out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(4))
    return out


def main():
    rnd, d = sys.argv[1], sys.argv[2]
    res = {}
    for leg, (grp, prefixes) in LEGS.items():
        if not os.path.exists(f"{d}/{rnd}_legs{grp}_fetch.md"):
            continue
        f = counter_sums(f"{d}/{rnd}_legs{grp}_fetch.md", "FETCH_SIZE")
        w = counter_sums(f"{d}/{rnd}_legs{grp}_write.md", "WRITE_SIZE")
        pick = lambda t: sum(v for k, v in t.items() if any(p in k for p in prefixes))
        rd, wr = 2 * pick(f) * 1024 / CALLS, pick(w) * 1024 / CALLS
        res[leg] = {"hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                    "kernels": sorted(set(k for k in list(f) + list(w) if any(p in k for p in prefixes))),
                    "source": f"profiles/{rnd}_legs{grp}_fetch.md + profiles/{rnd}_legs{grp}_write.md (rocprofv3 --pmc passes of "
                              f"scripts/quick_legs.py {grp}: {CALLS} launches; FETCH_SIZE x 2 and WRITE_SIZE x 1 KB per MI355X_MICROARCH.md); "
                              "NOT measured by the bench run"}
    k2 = k2_leg(rnd, d)
    if k2:
        res["mash_distance"] = k2
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
