#!/usr/bin/env python3
"""profiles/traffic.json from the round's PMC summaries (tools/prof_r01.sh, tools/prof_cfg.sh): HBM bytes per scanned row =
(FETCH_SIZE x 2 [gfx950 reports half of a wide coalesced read stream: MI355X_MICROARCH.md, HBM] + WRITE_SIZE) KiB per
dispatch x 1024 / rows, summed over the kernels of one scan.  usage: make_traffic.py <tag>   (e.g. r04)"""
import json, os, re, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(5))
    return out


def entry(path, rows, pick, what):
    c = counters(path)
    ks = [k for k in c if pick(k)]
    f = sum(c[k].get("FETCH_SIZE", 0.0) for k in ks)
    w = sum(c[k].get("WRITE_SIZE", 0.0) for k in ks)
    if not ks or f == 0:
        return None
    return {"hbm_bytes_per_row": (2 * f + w) * 1024 / rows,
            "source": "profiles/%s: FETCH_SIZE %.1f KiB/dispatch x2 (gfx950 half-count) + WRITE_SIZE %.1f KiB/dispatch over %g rows, %s (%s)" % (
                os.path.basename(path), f, w, rows, " + ".join(k.split("sybl::")[-1].split("(")[0] for k in ks), what)}


rec = json.load(open(os.path.join(root, "traffic.json")))
jobs = [
    ("7_cols_strategy_2_packed", "%s_pmc.txt" % tag, 1e9, lambda k: "k_scan_packed<3, 2, 2" in k, "the headline, compact storage: 16 stored B/row"),
    ("3_cols_strategy_2_packed", "%s_cfg2_pmc.txt" % tag, 1e8, lambda k: "k_scan_packed<0, 1, 2" in k or "k_fold" in k, "config 2, compact storage: 9 stored B/row"),
    ("3_cols_strategy_4_packed", "%s_cfg5_pmc.txt" % tag, 1e9, lambda k: "k_scan_packed<0, 1, 1" in k, "config 5, compact storage: 10 stored B/row"),
    # (from round 6 the counting pass -- k_count_key -- runs in a prepared query's first scan only: a step's traffic is without it)
    ("2_cols_strategy_5_packed", "%s_cfg4_pmc.txt" % tag, 1e9,
     lambda k: any(x in k for x in ("k_emit_packed", "k_part_hist", "k_part_fix") + (("k_count_packed", "k_count_key") if tag < "r06" else ())),
     "config 4, compact storage: 6 stored B/row + 4 B of record written and read per value + the 525 MB bucket table written"
     + ("; the counting pass over the key column runs once per prepared query, not per step" if tag >= "r06" else "")),
]
for key, fn, rows, pick, what in jobs:
    p = os.path.join(root, fn)
    if os.path.exists(p):
        e = entry(p, rows, pick, what)
        if e:
            rec[key] = e
            print(key, round(e["hbm_bytes_per_row"], 3))
json.dump(rec, open(os.path.join(root, "traffic.json"), "w"), indent=1)
