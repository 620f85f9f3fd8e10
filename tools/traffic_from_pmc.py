#!/usr/bin/env python
"""profiles/traffic.json from two rocprofv3 PMC passes over `python bench.py ...` (FETCH_SIZE pass, WRITE_SIZE pass).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/traffic_from_pmc.py out/f/p_counter_collection.csv out/w/p_counter_collection.csv "<provenance note>" [algorithmic.json]

algorithmic.json = `python bench.py --dump-algorithmic algorithmic.json` of the same build: the ALGORITHMIC bytes per launch of every
GEMM / convolution class (kernel instantiation | grid), which the PMC classes are then set beside (`algorithmic_mbytes_per_launch`,
`hbm_over_algorithmic`; a split-K class's second pass is its own kernel and is not in the algorithmic figure).

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE counts 128-byte requests
as 64 bytes for wide coalesced streams (MI355X_MICROARCH.md §HBM), which is what these kernels issue (16 B per lane)."""
import collections
import csv
import json
import os
import sys

GROUPS = {"mma_pipe_body (gemm + conv3x3: mma_pipe_kernel / mma_lat_kernel)": ("mma_pipe_kernel", "mma_pipe_pair_kernel", "mma_lat_kernel", "mma_lat_pair_kernel", "mma_fat_kernel", "mma_kernel", "splitk_reduce_kernel"),
          "attn_fwd_kernel": ("attn_fwd_kernel",), "ff_fused_kernel": ("ff_fused_kernel",),
          "groupnorm": ("gn_stats", "gn_apply", "gn_fused"), "layernorm": ("layernorm_kernel",)}


def load(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for g, pats in GROUPS.items():
                if any(p in r["Kernel_Name"] for p in pats):
                    tot[g] += float(r["Counter_Value"])
                    key = (g, r["Dispatch_Id"])
                    if key not in seen and "splitk_reduce" not in r["Kernel_Name"]:   # the reduce pass belongs to its GEMM launch
                        seen.add(key)
                        n[g] += 1
    return tot, n


def by_grid(path, counter):
    """Counter totals and launch counts per (kernel name, grid size): the grid identifies the shape class of a launch."""
    tot, n = collections.Counter(), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            key = (name.split("(")[0][:70], r.get("Grid_Size", "?"))
            tot[key] += float(r["Counter_Value"])
            n[key] += 1
    return tot, n


def main():
    fpath, wpath, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    ft, fn = load(fpath, "FETCH_SIZE")
    wt, wn = load(wpath, "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from storygen_amd.build import source_hash
    out = {"note": note, "kernel_source_hash": source_hash(),
           "formula": "2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), averaged over every launch of the run", "kernels": {}}
    for g in GROUPS:
        if fn[g] == 0 or wn[g] == 0:
            continue
        fetch, write = ft[g] / fn[g] * 1024.0, wt[g] / wn[g] * 1024.0
        out["kernels"][g] = {"launches_fetch_pass": fn[g], "launches_write_pass": wn[g], "fetch_size_bytes_raw": round(fetch),
                             "write_size_bytes": round(write), "hbm_bytes_per_launch": round(2 * fetch + write)}
    # the ten (kernel, grid) classes that move the most bytes over the run: where a traffic reduction would have to come from
    gf, gn_ = by_grid(fpath, "FETCH_SIZE")
    gw, _ = by_grid(wpath, "WRITE_SIZE")
    rows = sorted(((2 * gf[k] + gw.get(k, 0.0)) * 1024.0, k) for k in gf)[::-1][:16]
    alg = {}
    if len(sys.argv) > 4:
        with open(sys.argv[4]) as f:
            alg = json.load(f)["classes"]
    out["top_traffic_classes"] = []
    for b, k in rows:
        e = {"kernel": k[0], "grid": k[1], "launches": gn_[k], "hbm_mbytes_per_launch": round(b / gn_[k] / 1e6, 2),
             "hbm_gbytes_total": round(b / 1e9, 3)}
        a = alg.get(f"{k[0]}|{k[1]}")
        if a is not None:
            e["algorithmic_mbytes_per_launch"] = round(a["algorithmic_bytes"] / a["launches"] / 1e6, 2)
            e["hbm_over_algorithmic"] = round(e["hbm_mbytes_per_launch"] / max(e["algorithmic_mbytes_per_launch"], 1e-9), 2)
            e["shapes"] = a["shapes"]
        out["top_traffic_classes"].append(e)
    if alg:     # the whole GEMM / convolution group: launch-weighted algorithmic bytes (same launches as `kernels` above, reduce passes excluded)
        tot_b = sum(a["algorithmic_bytes"] for a in alg.values())
        tot_n = sum(a["launches"] for a in alg.values())
        g = out["kernels"].get("mma_pipe_body (gemm + conv3x3: mma_pipe_kernel / mma_lat_kernel)")
        if g is not None and tot_n:
            g["algorithmic_bytes_per_launch"] = round(tot_b / tot_n)
            g["hbm_over_algorithmic"] = round(g["hbm_bytes_per_launch"] / (tot_b / tot_n), 2)
    out["total_hbm_gbytes_all_kernels"] = round(sum((2 * gf[k] + gw.get(k, 0.0)) * 1024.0 for k in gf) / 1e9, 3)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
