"""profiles/rNN_pmc_traffic.json from two rocprofv3 PMC passes of the eager training step (one with --pmc FETCH_SIZE, one
with --pmc WRITE_SIZE; TCC has too few slots for both).  usage:
    python scripts/pmc_traffic.py out.json <fetch counter_collection.csv> <write counter_collection.csv> [bench.py options of the passes]
Units: KB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md "HBM"); Infinity
Cache hits are included.  The file records the hash of the kernel sources it was collected on (bench.py refuses a
number from other sources) and the kernel symbols behind each family."""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import bench_config_key, kernel_source_sha, parse  # noqa: E402

FAMILIES = {  # bench.py family name -> substrings of the kernel symbols it covers
    "conv_mfma_kernel": ("conv_pp_kernel", "persist11conv_kernel", "conv_mfma_kernel"),
    "wgrad_mfma_kernel": ("wgrad_dma_kernel", "wgrad_group_kernel"),
    "adam_ema_kernel": ("adam_ema_kernel", "adam_fused_kernel", "adam_proj_fused_kernel", "wgrad_mfma_kernelIDF16bLi128ELi128ELb1"),
}


STEP_BEGIN = ("step_prologue_kernel", "dg_zero_multi")   # the launch that opens a training step: counts the steps in a trace


def load(path, counter):
    tot, n, names = collections.Counter(), collections.Counter(), collections.defaultdict(set)
    with open(path, newline="") as f:
        rows = [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    prev_begin = False
    if True:
        for r in rows:
            tot["_all"] += float(r["Counter_Value"])
            n["_all"] += 1
            # a step opens with ONE run of step-begin launches (eager steps on a host batch issue fetch_reals as a second
            # prologue launch right behind the zero-fill / draws one: round 6)
            begin = any(sb in r["Kernel_Name"] for sb in STEP_BEGIN)
            if begin and not prev_begin:
                n["_steps"] += 1
            prev_begin = begin
            for fam, subs in FAMILIES.items():
                if any(s in r["Kernel_Name"] for s in subs):
                    tot[fam] += float(r["Counter_Value"])
                    n[fam] += 1
                    names[fam].add(r["Kernel_Name"][:120])
    return tot, n, names


def main():
    out, fetch_csv, write_csv = sys.argv[1:4]
    bench_args = parse(sys.argv[4:])  # the bench.py options the two PMC passes were run with (none = the default workload)
    fk, fn, names = load(fetch_csv, "FETCH_SIZE")
    wk, wn, _ = load(write_csv, "WRITE_SIZE")
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        sha = ""
    res = {"_note": __doc__.strip().split("usage")[0].strip() + " bytes_per_launch = (2*FETCH_KB + WRITE_KB)*1024 / launches.",
           "kernel_source_sha": kernel_source_sha(), "config": bench_config_key(bench_args, bench_args.arch or "none"),
           "git_sha": sha or "(collected on the GPU box: no .git there)", "kernels": {}}
    # the traces hold `steps` whole training steps (warm-up included: --steps 4 --warmup 2 = 6): per-step figures divide by it
    assert fn["_steps"] == wn["_steps"] and fn["_steps"] > 0, (fn["_steps"], wn["_steps"])
    steps = fn["_steps"]
    res["steps"] = steps
    for fam in FAMILIES:
        if fn[fam] == 0:
            continue
        assert fn[fam] == wn[fam], (fam, fn[fam], wn[fam])
        res["kernels"][fam] = {"fetch_kb_raw": fk[fam], "write_kb": wk[fam], "launches": fn[fam],
                               "bytes_per_launch": (2 * fk[fam] + wk[fam]) * 1024 / fn[fam],
                               "bytes_per_step": (2 * fk[fam] + wk[fam]) * 1024 / steps, "symbols": sorted(names[fam])}
    # every kernel of the trace, same correction (the doubling is calibrated for 16-byte-per-lane streams only: kernels with
    # narrower reads are over-counted by it - an upper bound on the step's HBM-side traffic)
    res["all_kernels"] = {"fetch_kb_raw": fk["_all"], "write_kb": wk["_all"], "launches": fn["_all"],
                          "bytes_per_step": (2 * fk["_all"] + wk["_all"]) * 1024 / steps,
                          "bytes_per_step_fetch_not_doubled": (fk["_all"] + wk["_all"]) * 1024 / steps}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"steps": steps, "per_launch": {k: round(v["bytes_per_launch"]) for k, v in res["kernels"].items()},
                      "per_step": {k: round(v["bytes_per_step"]) for k, v in res["kernels"].items()},
                      "step_total": round(res["all_kernels"]["bytes_per_step"])}))


if __name__ == "__main__":
    main()
