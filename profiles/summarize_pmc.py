"""Summarise rocprofv3 PMC passes into the per-kernel HBM-traffic JSON that bench.py reads.

usage: python profiles/summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>

Each pass is `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> -- python3 bench.py ...` (counters
in their own runs, MI355X_MICROARCH.md "HBM traffic").  Units: KiB per dispatch.  gfx950 correction: FETCH_SIZE reports
half of the bytes of a wide coalesced stream (checked on bn_add_relu, a pure dwordx4 stream) -> doubled; WRITE_SIZE exact.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
                name = re.sub(r"^void ", "", name).split("(")[0]
                acc[name].append(float(row["Counter_Value"]))
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    what = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (B=256, 1xMI355X)"
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fr = sum(fetch.get(k, [0.0])) / max(len(fetch.get(k, [])), 1)
        wr = sum(write.get(k, [0.0])) / max(len(write.get(k, [])), 1)
        kernels[k] = {"dispatches": len(fetch.get(k, [])) or len(write.get(k, [])),
                      "fetch_KiB_raw_mean": round(fr, 1), "write_KiB_mean": round(wr, 1),
                      "hbm_bytes_per_launch_corrected": int((2.0 * fr + wr) * 1024)}
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE over `" + what + "`.  KiB per dispatch, "
            "mean over the dispatches of a kernel (Generator-side launches carry B clips, Detector-side 2B).  FETCH doubled (gfx950 "
            "correction), WRITE exact.  bytes_per_step = sum over kernels of dispatches x mean bytes / steps run (3 = 1 warm-up + 2).")
    steps = 3.0
    total = sum(v["dispatches"] * v["hbm_bytes_per_launch_corrected"] for v in kernels.values()) / steps
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha          # digest of csrc/*.hip + *.hpp: bench.py drops a summary taken on other sources
    json.dump({"note": note, "kernel_source_sha": kernel_source_sha(), "hbm_bytes_per_step": int(total), "kernels": kernels},
              open(out, "w"), indent=1)
    print(f"{len(kernels)} kernels -> {out}")


if __name__ == "__main__":
    main()
