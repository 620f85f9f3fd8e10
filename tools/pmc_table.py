#!/usr/bin/env python
"""Per-dispatch PMC table from one rocprofv3 SQ pass over tools/pmc_kernels.py (development tool; profiles/*pmc*).

    C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d out -o p -- python tools/pmc_kernels.py attn conv gemm
    python tools/pmc_table.py out/p_counter_collection.csv out/p_kernel_trace.csv

Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles and disjoint
(wait_any + wait_inst + active ~ 1); SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, so mfma_util = busy / (GRBM_GUI_ACTIVE / 8 XCDs
x 1024 SIMDs); clk = GRBM_GUI_ACTIVE / 8 / duration.  The last of the repeated launches of each kernel configuration is shown.
A second pass with C2="SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" adds the
instruction mix (counts per wave instruction) and the LDS-issue wait share."""
import collections
import csv
import sys


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(names)


def main():
    cpath, tpath = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
    disp = collections.OrderedDict()
    with open(cpath) as f:
        for r in csv.DictReader(f):
            d = disp.setdefault(r["Dispatch_Id"], dict(name=r["Kernel_Name"], grid=r.get("Grid_Size", "?"), c={}))
            d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if "Start_Timestamp" in r and r["Start_Timestamp"]:
                d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if tpath:
        with open(tpath) as f:
            for r in csv.DictReader(f):
                did = col(r, "Dispatch_Id")
                if did in disp:
                    disp[did]["ns"] = float(col(r, "End_Timestamp")) - float(col(r, "Start_Timestamp"))
    last = collections.OrderedDict()
    for d in disp.values():
        if any(k in d["name"] for k in ("mma_", "attn_", "splitk", "gn_", "layernorm", "ff_fused")):
            last[(d["name"], d["grid"])] = d
    for (name, grid), d in last.items():
        c = d["c"]
        wave = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        ns = d.get("ns", 0.0)
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
        print(f"{short:64s} grid={int(float(grid)):8d} t={ns / 1e3:7.1f}us clk={gui / max(ns, 1.0):4.2f}GHz "
              f"mfma_util={100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(gui * 1024, 1.0):5.1f}% "
              f"wait_any={100 * c.get('SQ_WAIT_ANY', 0.0) / wave:5.1f}% wait_inst={100 * c.get('SQ_WAIT_INST_ANY', 0.0) / wave:5.1f}% "
              f"active={100 * c.get('SQ_ACTIVE_INST_ANY', 0.0) / wave:5.1f}% insts_valu={c.get('SQ_INSTS_VALU', 0.0):.2e} "
              f"lds_conflict={100 * c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0):5.2f}%"
              + ("" if "SQ_INSTS_MFMA" not in c else
                 f" | insts: mfma={c.get('SQ_INSTS_MFMA', 0.0):.2e} valu={c.get('SQ_INSTS_VALU', 0.0):.2e} salu={c.get('SQ_INSTS_SALU', 0.0):.2e} "
                 f"vmem={c.get('SQ_INSTS_VMEM', 0.0):.2e} lds={c.get('SQ_INSTS_LDS', 0.0):.2e} "
                 f"wait_inst_lds={100 * c.get('SQ_WAIT_INST_LDS', 0.0) / wave:5.1f}% valu_per_mfma={c.get('SQ_INSTS_VALU', 0.0) / max(c.get('SQ_INSTS_MFMA', 0.0), 1.0):5.2f}"))


if __name__ == "__main__":
    main()
