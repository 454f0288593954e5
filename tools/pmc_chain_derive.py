"""tools/pmc_chain_derive.py DIR [GHZ] -- per-kernel reading of the counter passes of tools/pmc_chain_batch.sh.

Counters are per launch, summed over the 8 XCDs.  Under counter collection GRBM_GUI_ACTIVE carries ~10 us of collection overhead per launch, so
the launch duration is taken from the kernel trace of the same pass (it matches the unprofiled run) and converted with GHZ (default 2.05: what
GRBM_GUI_ACTIVE / duration gives for the 570 us period warp, where the overhead does not matter).  A wave instruction occupies its 16-lane SIMD
for four cycles and SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count those four-cycle slots: fraction = count x 4 / (cycles x 1,024 SIMDs);
*_sum counters of TD / TCP are summed over 256 CUs; *_avr are per-instance averages."""
import collections, csv, glob, re, sys

root = sys.argv[1]
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.05
name = lambda s: (lambda m: f"{m.group(1)}<{m.group(2)}>" if m else None)(re.search(r"(flow_\w+|blur_\w+)<([^>]*)>", s))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name(r["Kernel_Name"]): acc[name(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name(r["Kernel_Name"]): dur[name(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# per launch; cycles = kernel-trace duration x {ghz} GHz")
print("%-42s %6s %6s %6s %6s %6s %6s %6s %6s %6s %6s %7s %7s %8s" % ("kernel", "us", "waves", "VALU/w", "SALU/w", "VMEM/w", "occ/SIMD", "valu%", "wait%",
                                                                          "td%", "tcc%", "L2hit%", "L2->L1MB", "TB/s"))
for k, d in sorted(acc.items()):
    g = lambda c: sum(d[c]) / len(d[c]) if d.get(c) else float("nan")
    us = sum(dur[k]) / len(dur[k])
    cyc = us * 1e3 * ghz
    w = g("SQ_WAVES")
    mb = g("TCP_TCC_READ_REQ_sum") * 64 / 1e6
    print("%-42s %6.1f %6.0f %6.0f %6.0f %6.1f %6.2f %6.1f %6.1f %6.1f %6.1f %7.1f %7.1f %8.2f" % (
        k[:42], us, w, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_WAVE_CYCLES") * 4 / (cyc * 1024),
        100 * g("SQ_ACTIVE_INST_VALU") * 4 / (cyc * 1024), 100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        100 * g("TD_TD_BUSY_sum") / (cyc * 256), 100 * g("TCC_BUSY_avr") / cyc, 100 * g("TCC_HIT_sum") / g("TCC_REQ_sum"), mb, mb / us))
