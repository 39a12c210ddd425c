"""Copy the judged round-6 rocprofv3 summaries from gpurun_out/prof_r6 into profiles/ (tracked)."""
import csv, collections, statistics, re, glob, os, shutil, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r6")
out = os.path.join(ROOT, "profiles")
def kname(n):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]
for d, name in (("solo_head", "r6_solo_kernel_stats.csv"), ("solo_tail", "r6_solo_tail_kernel_stats.csv"),
                ("bench", "r6_bench_kernel_stats.csv"),
                ("chip_full", "r6_chip_full_kernel_stats.csv"), ("ring", "r6_reference_ring_kernel_stats.csv")):
    ks = glob.glob(os.path.join(src, d, "**/*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(out, name))
def counter_files(d):
    return glob.glob(os.path.join(src, d, "**/*counter_collection.csv*"), recursive=True)
old = {}
tj = os.path.join(out, "k1_traffic.json")
if os.path.exists(tj):
    old = json.load(open(tj))
traffic, lines = {k: v for k, v in old.items() if not k.endswith("_lean1024")}, []   # (the 1024-thread lean kernel is gone)
def collect(label, suffix, keys):
    """keys: (json key for the scatter kernel, json key for the stencil kernel)"""
    k1, k3 = {}, {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = counter_files("%s_%s" % (cname, suffix))
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == cname:
                acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            live = [x for x in v if x > 64] or [0.0]
            lines.append("%-34s %-11s %-44s dispatches %5d  live %5d  median %12.1f KB  mean %12.1f KB" %
                         (label, cname, k, len(v), len(live), statistics.median(live), sum(live) / len(live)))
            f = "fetch_kb" if cname == "FETCH_SIZE" else "write_kb"
            if k.startswith("k_bin_warp_scatter") and "<true" in k:
                k1[f] = statistics.median(live); k1["kernel"] = k
            if k.startswith("k_stencil_binned"):
                k3[f] = statistics.median(live); k3["kernel"] = k
    if len(k1) >= 3 and keys[0]:
        traffic[keys[0]] = k1
    if len(k3) >= 3 and keys[1]:
        traffic[keys[1]] = k3
collect("346x260 lean 512 + tail update", "346x260_lean512", ("346x260x3_lean512", "346x260x3_stencil_tail"))
collect("346x260 head update 1024", "346x260_head1024", ("346x260x3_head1024", "346x260x3_stencil_head"))
collect("640x480 co-scheduled shape", "640x480", ("640x480x3", "640x480x3_stencil_tail"))
collect("1280x720 co-scheduled shape", "1280x720", ("1280x720x3", "1280x720x3_stencil_tail"))
traffic["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (scripts/profile_r6.sh; keys without a variant suffix and "
                     "the *_head_split* keys: round 3's scripts/profile_r3.sh), median over the live launches of the named kernel of one "
                     "cold 1M-event slice per geometry; FETCH_SIZE is doubled by the reader (gfx950)")
json.dump(traffic, open(tj, "w"), indent=1)
open(os.path.join(out, "r6_pmc_hbm_traffic.txt"), "w").write(
    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) over one cold 1M-event\n"
    "slice per geometry and kernel variant (scripts/run_once.py 1 <options>; 640x480 and 1280x720: first 300 iterations).\n"
    "KB per dispatch; 'live' excludes the early-exit launches after convergence.  On gfx950 FETCH_SIZE under-reports wide\n"
    "coalesced reads by 2x (MI355X_MICROARCH.md, HBM section): double it before comparing with byte counts.\n\n" + "\n".join(lines) + "\n")
for d, name in (("sq_720", "r6_pmc_sq_720p.txt"), ("sq_346", "r6_pmc_sq_issue.txt")):
    fs = counter_files(d)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = kname(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    rows = ["rocprofv3 --kernel-trace --pmc SQ_* (own pass) over one cold slice (%s), per launch, summed over the device." % d,
            "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.", ""]
    for k in sorted(agg, key=lambda k: -agg[k]["SQ_BUSY_CYCLES"])[:6]:
        a, n = agg[k], len(seen[k])
        wc = a["SQ_WAVE_CYCLES"] or 1.0
        rows.append("%-44s launches %4d  waves %6.0f  VALU insts/wave %6.0f  wave-cycles: active %4.1f%% (VALU %4.1f%%)  "
                    "wait(s_waitcnt/barrier) %4.1f%%  issue-stall %4.1f%%" %
                    (k, n, a["SQ_WAVES"] / n, a["SQ_INSTS_VALU"] / max(a["SQ_WAVES"], 1), 100 * a["SQ_ACTIVE_INST_ANY"] / wc,
                     100 * a["SQ_ACTIVE_INST_VALU"] / wc, 100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc))
    open(os.path.join(out, name), "w").write("\n".join(rows) + "\n")
    print("\n".join(rows))
    # vector instructions per wave of the loop kernels, machine readable: bench.py prices the event-list stencil kernel's
    # COMPUTE-side bound with it (waves x instructions / the chip's issue rate)
    vj = os.path.join(out, "r6_valu.json")
    vd = json.load(open(vj)) if os.path.exists(vj) else {}
    vd[{"sq_720": "1280x720", "sq_346": "346x260"}[d]] = {
        k: {"valu_per_wave": agg[k]["SQ_INSTS_VALU"] / max(agg[k]["SQ_WAVES"], 1), "waves_per_launch": agg[k]["SQ_WAVES"] / len(seen[k]),
            "launches": len(seen[k]), "valu_active_share_of_wave_cycles": agg[k]["SQ_ACTIVE_INST_VALU"] / (agg[k]["SQ_WAVE_CYCLES"] or 1.0)}
        for k in agg if k.startswith("k_stencil_binned") or k.startswith("k_bin_warp_scatter")}
    vd["source"] = "rocprofv3 --kernel-trace --pmc SQ_* (scripts/profile_r6.sh), one cold slice per geometry, co-scheduled kernel variants"
    json.dump(vd, open(vj, "w"), indent=1)
for log, name in (("bench.log", "r6_bench_under_rocprof.json"),):
    bl = os.path.join(src, log)
    if os.path.exists(bl):
        for ln in open(bl):
            if ln.startswith('{"metric"'):
                open(os.path.join(out, name), "w").write(ln)
for log, name in (("chip_full.log", "r6_chip_full_under_rocprof.txt"), ("ring.log", "r6_reference_ring_under_rocprof.txt")):
    bl = os.path.join(src, log)
    if os.path.exists(bl):
        shutil.copy(bl, os.path.join(out, name))
print(open(os.path.join(out, "r6_pmc_hbm_traffic.txt")).read())
for f in ("r6_solo_kernel_stats.csv", "r6_solo_tail_kernel_stats.csv", "r6_chip_full_kernel_stats.csv", "r6_reference_ring_kernel_stats.csv"):
    p = os.path.join(out, f)
    if os.path.exists(p):
        print(f); print("".join(open(p).readlines()[:6]))
