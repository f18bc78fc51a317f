"""Condense a tools/profile_round.sh output directory into the small files kept under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the default bench
  profiles/<tag>_counters.json      per-launch averages of the PMC passes (HBM traffic with the
                                    gfx950 FETCH_SIZE correction, VALU / LDS instruction counts)
usage: python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> [chunk]"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys


def counter_avgs(d):
    # group by grid size and keep the k-loop's launches (the most frequent grid): k_crt_acc is also
    # launched, with other sizes, by the key generation that precedes the timed batch
    res = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            for k in ("k_extprod", "k_crt_lean", "k_crt_acc", "k_init", "k_final"):
                if k in name:
                    res[k][r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, grids in res.items():
        cs = max(grids.values(), key=lambda g: max(len(v) for v in g.values()))
        out[k] = {c: (sum(v) / len(v), len(v)) for c, v in cs.items()}
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(prof, "%s_kernel_stats.csv" % tag))
    log = os.path.join(src, "bench_stats.log")
    bench_line = None
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(prof, "%s_bench_under_rocprof.json" % tag), "w").write(lines[-1])
            bench_line = json.loads(lines[-1])
    out = {"note": "rocprofv3 --pmc passes (one counter group per run, --kernel-trace only), "
                   "bench.py --lanes 1 --chunk %d --batch %d --steps 1 --warmup 0 = one chunk of %d bootstraps, " % (chunk, chunk, chunk) +
                   "Params(1024); per-launch averages. FETCH_SIZE / WRITE_SIZE are reported in KB; "
                   "on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads "
                   "(MI355X_MICROARCH.md, HBM section): traffic = (2 FETCH_SIZE + WRITE_SIZE) * 1024 B.",
           "chunk": chunk, "kernels": {}}
    sys.path.insert(0, root)
    import sgfhe_jl_amd
    # bench.py quotes these numbers only beside a library with this sgfhe_build_id(): the id the
    # profiled run itself reported, else the hash of the sources here
    out["build_id"] = (bench_line or {}).get("config", {}).get("build_id") or sgfhe_jl_amd.source_hash()
    def kernel_avgs(path):
        avg = {}
        for r in csv.DictReader(open(path)):
            for k in ("k_extprod", "k_crt_lean"):
                if "::" + k + "<" in r["Name"]:
                    avg[k] = float(r["AverageNs"]) / 1e3
        return avg
    if stats:   # rocprofv3 --kernel-trace --stats averages of the two k-loop kernels in the default
                # schedule (two lanes: durations under co-execution)
        out["rocprof_avg_us"] = kernel_avgs(stats[0])
        # device wall time of one k-loop iteration of one chunk, from the dispatch time stamps of
        # the same trace: span of the k-loop's dispatches / chunk-iterations in it
        trace = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)
        if trace and bench_line:
            t0, t1, n_ext = None, None, 0
            for r in csv.DictReader(open(trace[0])):
                name = r.get("Kernel_Name", "")
                if "k_extprod" in name or "k_crt_lean" in name:
                    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                    t0 = a if t0 is None else min(t0, a)
                    t1 = b if t1 is None else max(t1, b)
                    n_ext += "k_extprod" in name
            if n_ext:
                out["rocprof_iter_us"] = (t1 - t0) / 1e3 / n_ext
                out["rocprof_iter_note"] = ("(last end - first start) of the %d k-loop dispatches of the "
                                            "profiled run / k_extprod launches" % (2 * n_ext))
    alone = glob.glob(os.path.join(src, "stats_alone", "**", "*kernel_stats.csv"), recursive=True)
    if alone:   # the same chunks on one lane: each kernel alone on the device
        out["rocprof_alone_avg_us"] = kernel_avgs(alone[0])
        shutil.copy(alone[0], os.path.join(prof, "%s_kernel_stats_one_lane.csv" % tag))
    try:
        out["valu_mix"] = json.loads(subprocess.check_output(
            [sys.executable, os.path.join(root, "tools", "valu_mix.py")]).decode())
    except Exception as exc:                   # no hipcc on this box: the mix is optional
        out["valu_mix"] = None
        out["valu_mix_error"] = str(exc)
    merged = collections.defaultdict(dict)
    for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ", "pmc_SQW", "pmc_SQL"):
        for k, cs in counter_avgs(os.path.join(src, sub)).items():
            for c, (avg, n) in cs.items():
                merged[k][c] = avg
                merged[k]["launches"] = n
    for k, cs in merged.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            cs["traffic_bytes_per_launch"] = (2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in cs and "SQ_WAVES" in cs and cs["SQ_WAVES"]:
            cs["valu_insts_per_wave"] = cs["SQ_INSTS_VALU"] / cs["SQ_WAVES"]
        out["kernels"][k] = cs
    json.dump(out, open(os.path.join(prof, "%s_counters.json" % tag), "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
