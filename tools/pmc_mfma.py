#!/usr/bin/env python
"""Per-kernel MFMA utilisation from a rocprofv3 PMC pass (counter_collection.csv of
`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES -- python3 bench.py ...`).

    python tools/pmc_mfma.py <p_counter_collection.csv> [out.txt]

Per kernel (summed over its dispatches): duration, MFMA-busy cycles, and
    mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (duration x SIMDs x clock)   -- share of the run time the matrix pipes are busy,
                    at the 2.4 GHz peak clock (the chip clocks lower under load, so this is a lower bound of the busy share
                    and exactly "fraction of the peak MFMA issue rate" -- /opt/skills/guides/MI355X_MICROARCH.md:
                    SQ_VALU_MFMA_BUSY_CYCLES = 32 x N for v_mfma_f32_32x32x16_bf16, cycles, summed over SIMDs)
    valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA     (non-matrix vector instructions issued per matrix instruction;
                    SQ_INSTS_VALU counts the MFMAs too on gfx950, so 1.0 = nothing but MFMAs)
Counter semantics are those of this rocprofv3 build on gfx950 (no derived-metric section ships for it): treat the
absolute numbers as indicative, the ratios between kernels and between rounds as the evidence."""
import collections
import csv
import re
import sys

SIMDS, CLOCK = 256 * 4, 2.4e9


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "")[:100]


def main():
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            d = (k, r["Dispatch_Id"])
            if d not in seen:
                seen.add(d)
                rows[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                rows[k]["_n"] += 1
    tot = sum(v["_ns"] for v in rows.values())
    out = [f"{'kernel':100s} {'calls':>5s} {'sum_ms':>8s} {'%time':>6s} {'mfma_busy':>9s} {'valu/mfma':>9s} {'mfma insts':>12s}"]
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["_ns"]):
        if v["_ns"] / tot < 0.002:
            continue
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["_ns"] * 1e-9 * CLOCK * SIMDS)
        nm = v.get("SQ_INSTS_MFMA", 0.0)
        ratio = v.get("SQ_INSTS_VALU", 0.0) / nm if nm else float("nan")
        out.append(f"{k:100s} {int(v['_n']):5d} {v['_ns'] / 1e6:8.3f} {100 * v['_ns'] / tot:6.2f} {busy:9.3f} {ratio:9.2f} {nm:12.0f}")
    txt = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
