"""HBM-side traffic per kernel launch from two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
usage: python tools/pmc_traffic.py <dir of the --pmc FETCH_SIZE run> <dir of the --pmc WRITE_SIZE run> [build head] [workload]
Counter values are KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B read requests are
tallied at 64 B).  The last line (`#json {...}`) is what bench.py's roofline.traffic reads: it carries the hash of the
kernel sources the profiled build was made from, and bench.py quotes the figures only when that hash is the one it runs."""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0][:60]
        tot[n] += float(r["Counter_Value"])
        calls[n] += 1
    return tot, calls


def family(fetch, fc, write, wc, prefix):
    ks = [n for n in fetch if n.startswith(prefix)]
    nl = sum(fc[n] for n in ks)
    f_kb = sum(fetch[n] for n in ks) / max(nl, 1)
    w_kb = sum(write.get(n, 0.0) for n in ks) / max(sum(wc.get(n, 0) for n in ks), 1)
    return nl, f_kb, w_kb


def main():
    from focus_amd.build import source_hash
    fetch, fc = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, wc = per_kernel(sys.argv[2], "WRITE_SIZE")
    head = sys.argv[3] if len(sys.argv) > 3 else "unknown"
    wl = sys.argv[4] if len(sys.argv) > 4 else "orvit"
    print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and (separate run)  --pmc WRITE_SIZE  over  python3 bench.py --workload %s "
          "--steps 3 --warmup 1 --no-cpu-baseline --no-roofline   (build %s)" % (wl, head))
    print("# units: KB per launch; FETCH doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); WRITE as reported")
    print("%-62s %6s %16s %16s %16s" % ("kernel", "calls", "FETCH KB/launch", "x2 (gfx950)", "WRITE KB/launch"))
    rows = sorted(fetch, key=lambda n: -(2 * fetch[n] + write.get(n, 0.0)))
    for n in rows[:40]:
        c = max(fc[n], 1)
        print("%-62s %6d %16.0f %16.0f %16.0f" % (n, fc[n], fetch[n] / c, 2 * fetch[n] / c, write.get(n, 0.0) / max(wc.get(n, 1), 1)))
    rec = {"src": source_hash(), "head": head, "workload": wl, "executions": int(sys.argv[5]) if len(sys.argv) > 5 else 4,
           "families": {}}
    total = sum(2 * fetch[n] + write.get(n, 0.0) for n in fetch) * 1e3          # bytes over the whole trace
    rec["trace_total_bytes"] = round(total)
    for prefix in ("gemm_nt_ws_kernel", "gemm_nt8_kernel", "gemm_tn_ws_kernel", "traj_bwd_fused_kernel", "traj_dq_kernel", "traj_dkv_kernel",
                   "traj_delta_kernel", "traj_space_fwd_kernel", "slot_fwd_mfma_kernel", "slot_bwd_defer_kernel",
                   "slot_kv_grad_kernel", "slot_iter_kernel", "slot_frame_kernel"):
        nl, f_kb, w_kb = family(fetch, fc, write, wc, prefix)
        if nl:
            rec["families"][prefix] = {"launches": nl, "bytes_per_launch": round((2 * f_kb + w_kb) * 1e3)}
            print("%s (all instances): launches %d, HBM-side bytes per launch = 2*%.0f + %.0f KB = %.1f MB"
                  % (prefix, nl, f_kb, w_kb, (2 * f_kb + w_kb) / 1e3))
    print("#json " + json.dumps(rec))


if __name__ == "__main__":
    main()
