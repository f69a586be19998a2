"""HBM traffic per launch from two rocprofv3 PMC passes of `bench.py` (FETCH_SIZE and WRITE_SIZE in separate passes, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: the TCC block cannot hold both).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile
    python profiles/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01/v3_pmc_traffic.json

Units: both counters are reported in KiB.  Corrections: the guide's factor 2 for FETCH_SIZE holds for 16-B-per-lane
reads; this code base moves 8 B per lane (512 B per wavefront instruction), which the guide calls uncalibrated, so the
factor is calibrated on k_update, a pure streaming kernel with a known byte count (160 words read, 80 written per
(instance, interval)), at full width, and that factor is applied to the other kernels.
"""
import collections, csv, glob, json, sys

B, N = 8192, 40  # (B is re-derived from the widest k_update launch of the profiled run)
UPDATE_READ, UPDATE_WRITE = 160 * 8, 80 * 8  # bytes per (instance, interval)


def kname(n):
    """ltompc::k_eval<...>(args) / void ltompc::k_eval<ltompc::BoundsFixed<...> >(args) -> k_eval"""
    import re
    return re.sub(r"[<(].*", "", n.replace("void ", "").replace("ltompc::", "")).strip()


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[kname(r["Kernel_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return out


def merge_single_instance_sweeps(res):
    """bench.py's kernel class "riccati1" is k_riccati1 (one wavefront per instance) AND k_riccati1q (four, the narrowest launches):
    the class average over all launches of both goes where bench.py looks for it (`k_riccati1`.`traffic_avg_all_launches`)."""
    k1, kq = res["kernels"].get("k_riccati1"), res["kernels"].get("k_riccati1q")
    if k1 and kq and "traffic_avg_all_launches_own" not in k1:
        n1, nq = k1["all_launches"], kq["all_launches"]
        k1["traffic_avg_all_launches_own"] = k1["traffic_avg_all_launches"]
        k1["traffic_avg_all_launches"] = (k1["traffic_avg_all_launches"] * n1 + kq["traffic_avg_all_launches"] * nq) / (n1 + nq)
        k1["class_launches_with_k_riccati1q"] = n1 + nq


def main():
    global B
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    # instances of the profiled handle: bench.py splits the batch of a GPU into handles of B / parts instances (SplitMPC); a
    # full-width k_update launch has N * B threads
    B = max(g for g, _ in fetch["k_update"]) // N
    res = {"batch": B, "horizon": N, "unit": "bytes per full-width launch (grid = the whole batch)", "kernels": {}}
    full = {}
    for k in fetch:
        gmax = max(g for g, _ in fetch[k])
        fr = [v for g, v in fetch[k] if g == gmax]
        wr = [v for g, v in write.get(k, []) if g == gmax]
        if not wr:
            continue
        full[k] = (gmax, sum(fr) / len(fr) * 1024, sum(wr) / len(wr) * 1024, len(fr),
                   sum(v for _, v in fetch[k]) / len(fetch[k]) * 1024, sum(v for _, v in write[k]) / len(write[k]) * 1024, len(fetch[k]))
    cal_r = B * N * UPDATE_READ / full["k_update"][1]
    cal_w = B * N * UPDATE_WRITE / full["k_update"][2]
    res["calibration"] = {"kernel": "k_update", "fetch_factor": cal_r, "write_factor": cal_w,
                          "note": "known bytes / counter bytes at full width; guide: 2.0 for 16-B-per-lane reads, 1.0 for writes"}
    for k, (g, r, w, n, ra, wa, na) in sorted(full.items()):
        res["kernels"][k] = {"grid_threads": g, "launches": n, "fetch_raw": r, "write_raw": w,
                             "traffic": r * cal_r + w * cal_w,  # per full-width launch
                             "all_launches": na, "traffic_avg_all_launches": ra * cal_r + wa * cal_w}
    merge_single_instance_sweeps(res)
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:14s} grid {v['grid_threads']:8d} n={v['launches']:4d} fetch_raw {v['fetch_raw']/1e6:9.1f} MB write_raw {v['write_raw']/1e6:9.1f} MB traffic {v['traffic']/1e6:9.1f} MB")
    print("calibration", res["calibration"])


if __name__ == "__main__":
    main()
