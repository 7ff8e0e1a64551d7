"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md §HBM).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p --output-format csv -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write "conv_igemmg_kernel<64, 64, 2, 2, 1, 32>" \
        profiles/round1_pmc_hbm_traffic.txt profiles/roofline_traffic.json ALGORITHMIC_BYTES
Counter values are KB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so reads are doubled.
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")[:80]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fd, wd, kernel, txt, js, algo = sys.argv[1:7]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1, "
             "values in KB per launch (raw counter, average)",
             "gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced "
             "reads -> multiply FETCH by 2", ""]
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        lines.append(f"{k:82s} {fe[k][0]:12.0f} {wr.get(k, (0, 0))[0]:16.0f}  (n={fe[k][1]})")
    open(txt, "w").write("\n".join(lines) + "\n")
    key = [k for k in fe if kernel in k]
    assert len(key) == 1, (kernel, list(fe))
    f, w = fe[key[0]][0], wr[key[0]][0]
    out = {"kernel": key[0], "fetch_size_kb_raw": round(f), "write_size_kb": round(w), "fetch_correction": 2.0,
           "traffic_bytes_per_launch": int((2.0 * f + w) * 1024), "algorithmic_bytes_per_launch": int(algo),
           "launches_sampled": fe[key[0]][1],
           # every kernel of the same two passes (bytes per launch, averaged over its launches): bench.py looks its dominant kernel
           # up here when it is not the one above (two kernels of similar share swap places from run to run)
           "all_kernels": {k: int((2.0 * fe[k][0] + wr.get(k, (0.0, 0))[0]) * 1024) for k in fe},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1`, "
                     "MI355X; FETCH doubled per MI355X_MICROARCH.md §HBM (gfx950 counts 128-B requests as 64 B)"}
    json.dump(out, open(js, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
