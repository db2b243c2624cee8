"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (means per launch).

usage: python tools/pmc_summary.py out.txt out.json batch pass1_counter_collection.csv [pass2_counter_collection.csv ...]

out.json holds, per kernel, the mean counters per launch and the batch size they were taken at;
bench.py scales them to its own batch for the roofline's `traffic` and VALU-issue fields.

Derived lines: VALU issue utilisation assumes one wave-instruction occupies a SIMD for 4 cycles
(64 lanes over 16-lane ALUs) on 1024 SIMDs at 2.4 GHz; the opcodes gfx950 issues at twice that rate
(32-bit add / logic / shift right / move, f32 add / mul / fma: profiles/r02_valu_opcode_rates.txt)
push the figure above 100 % for a kernel made of them (the brute-force scan).  FETCH_SIZE / WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE counts half the bytes of coalesced reads (128-byte requests tallied as
64 bytes), for 4-byte-per-lane loads as for 16-byte ones, and WRITE_SIZE is exact:
tools/fetch_calib.hip measures both on a 512 MiB buffer (profiles/r01_fetch_calibration.txt).
The summary line therefore doubles FETCH_SIZE; the raw counter values are listed as collected.
"""
import csv
import sys
from collections import defaultdict


def main():
    out, out_json, batch, files = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
    import json
    js = {}
    vals = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(list)
    seen = set()
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"].split("(")[0]
                c = r["Counter_Name"]
                vals[k][c] += float(r["Counter_Value"])
                cnt[k][c] += 1
                key = (f, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines = ["# rocprofv3 --pmc, means per launch; durations are from the counter runs (slower clocks than un-profiled runs)"]
    for k in sorted(vals):
        if not (k.startswith("k_") or "k_" in k):
            continue
        d = sum(dur[k]) / len(dur[k])
        lines.append("%s: launches/pass=%d mean_ns=%.0f" % (k, len(dur[k]) // max(1, len(files)), d))
        m = {c: vals[k][c] / cnt[k][c] for c in vals[k]}
        js[k] = {"batch": batch, "mean_ns_profiled": d, "counters_per_launch": m}
        for c in sorted(m):
            lines.append("    %-28s %16.0f" % (c, m[c]))
        if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"]:
            w = m["SQ_WAVES"]
            util = m["SQ_INSTS_VALU"] * 4 / (1024 * 2.4 * d)
            lines.append("    VALU issue utilisation %.1f %%; per wave: valu %.0f salu %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f" % (
                100 * util, m["SQ_INSTS_VALU"] / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w,
                m.get("SQ_INSTS_VMEM_RD", 0) / w, m.get("SQ_INSTS_VMEM_WR", 0) / w))
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            lines.append("    memory-side traffic per launch: fetch %.1f MB (2 x FETCH_SIZE, see tools/fetch_calib.hip), write %.1f MB" % (
                2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6, m.get("WRITE_SIZE", 0) * 1024 / 1e6))
        if "SQ_LDS_IDX_ACTIVE" in m:
            lines.append("    LDS busy %.1f %% of CU-cycles (256 CUs)" % (100 * m["SQ_LDS_IDX_ACTIVE"] / (256 * 2.4 * d)))
    open(out, "w").write("\n".join(lines) + "\n")
    json.dump(js, open(out_json, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
