#!/usr/bin/env python3
"""profiles/traffic.json from rocprofv3 --pmc passes (tools/pmc.sh): HBM-side bytes and VALU instructions per launch of the
dominant kernel of each bench config, stamped with the id of the build they were measured on (vrt_build_id).

    python tools/make_traffic.py <build_id> <label> name=pmc_dir [name=pmc_dir ...]

FETCH_SIZE is reported by rocprofv3 in KB as 64 B x (L2 read requests to the fabric).  A streaming read issues 128-byte
requests (tools/probes/fetch_calib.cpp: 2 GiB streamed -> 1.0 GiB reported; MI355X_MICROARCH.md section HBM), so it is doubled
as the guide prescribes; for the isolated 4- and 8-byte reads of a walk (one request per read, reported as 64 B) the doubled
figure is the upper bound -- a whole 128-byte line per miss -- and `fetch_bytes_raw` the lower one.  WRITE_SIZE as read."""
import csv, glob, json, os, sys, collections


def kernel_means(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def pick(means, prefix, instr_arg):
    """The non-instrumented instantiation: template argument number `instr_arg` (None: not a template) is false."""
    for k, v in means.items():
        if k.replace("void ", "").startswith(prefix):
            if instr_arg is not None:
                args = [a.strip() for a in k.split("<")[1].rstrip(">").split(",")]
                if args[instr_arg] == "true":
                    continue
            return k, v
    return None, None


def entry(v):
    fetch_raw = v.get("FETCH_SIZE", 0.0) * 1024.0
    write = v.get("WRITE_SIZE", 0.0) * 1024.0
    e = {"fetch_bytes_raw": int(fetch_raw), "fetch_bytes_per_launch": int(2 * fetch_raw), "write_bytes_per_launch": int(write),
         "bytes_per_launch": int(2 * fetch_raw + write), "valu_wave_insts_per_launch": int(v.get("SQ_INSTS_VALU", 0)),
         "valu_issue_units_per_launch": int(v.get("SQ_ACTIVE_INST_VALU", 0))}
    if v.get("SQ_ACTIVE_INST_VALU"):
        e["lane_use"] = round(v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4)
    if v.get("SQ_WAVE_CYCLES"):
        e["wave_time_waiting"] = round(v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
    if (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
        e["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
    return e


def main():
    build_id, label = sys.argv[1], sys.argv[2]
    out = {"build_id": build_id, "label": label,
           "source": "rocprofv3 --kernel-trace --pmc <one group per pass> over tools/bench_scenes.py <config> with VRT_OVERLAP=0 (tools/pmc.sh); "
                     "mean per dispatch of the non-instrumented kernel",
           "correction": "FETCH_SIZE (KB) x 1024 x 2: 128-byte requests tallied at 64 B (calibrated with tools/probes/fetch_calib.cpp); WRITE_SIZE (KB) x 1024",
           "kernels": {}}
    for arg in sys.argv[3:]:
        name, d = arg.split("=", 1)
        m = kernel_means(d)
        ent = {}
        for short, prefix, ia in (("k_render_pool", "vrt::k_render_pool<", 1), ("k_render_pool_restir", "vrt::k_render_pool_restir<", 1),
                                  ("k_render", "vrt::k_render<", 2), ("k_gris", "vrt::k_gris<", 1),
                                  ("k_gris_prepare", "vrt::k_gris_prepare", None), ("k_temporal", "vrt::k_temporal", None)):
            k, v = pick(m, prefix, ia)
            if not v and short == "k_render_pool":   # the dense-grid variants of the same kernel (k_render_pool_dense12 / _dense<G, INSTR, CULL>)
                k, v = pick(m, "vrt::k_render_pool_dense12<", 1)
                if not v:
                    k, v = pick(m, "vrt::k_render_pool_dense<", 1)
            if v and short == "k_gris":
                # the spatial-reuse pass runs as two kernels (template argument 3 = 1, 2: vrt_restir.h): one entry, their counters summed
                # (round 3: behind a third, k_gris_classify<INSTR>, that hands them their masks of accepted and of live taps)
                halves = [(kk, vv) for kk, vv in m.items() if kk.replace("void ", "").startswith(prefix)
                          and [a_.strip() for a_ in kk.split("<")[1].rstrip(">").split(",")][ia] != "true"]
                halves += [(kk, vv) for kk, vv in m.items() if kk.replace("void ", "").startswith("vrt::k_gris_classify<false>")]
                # round 4: the first of the two is k_gris_first<G, INSTR> (wave-level schedule), and the per-pixel records' pass counts too
                halves += [(kk, vv) for kk, vv in m.items() if kk.replace("void ", "").startswith("vrt::k_gris_first<") and kk.rstrip(">").endswith("false")]
                halves += [(kk, vv) for kk, vv in m.items() if kk.replace("void ", "").startswith("vrt::k_gris_prepare")]
                if len(halves) > 1:
                    names = sorted(kk for kk, _ in halves)
                    summed = {}
                    for _, vv in halves:
                        for c_, x in vv.items():
                            summed[c_] = summed.get(c_, 0.0) + x
                    k, v = " + ".join(names), summed
            if v:
                ent[short] = dict(entry(v), kernel=k)
        out["kernels"][name] = ent
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps({n: {k: (e["bytes_per_launch"], e["valu_wave_insts_per_launch"], e.get("lane_use")) for k, e in v.items()} for n, v in out["kernels"].items()}))


if __name__ == "__main__":
    main()
