"""tools/pmc_instr_per_wave.py DIR... FILTER -- instructions per wave of the kernels whose name contains FILTER, from rocprofv3 --pmc
counter_collection CSVs (SQ_WAVES, SQ_INSTS_*, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE spread over one or more passes)."""
import collections, csv, glob, sys
*dirs, filt = sys.argv[1:]
d = collections.defaultdict(list)
for root in dirs:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if filt in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
w = m.get("SQ_WAVES", 1.0)
print("  waves per launch %d;  per wave: VALU %.0f  SALU %.0f  LDS %.1f  vector loads %.1f  vector stores %.1f" %
      (w, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_VMEM_RD", 0) / w, m.get("SQ_INSTS_VMEM_WR", 0) / w))
if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
    print("  SQ_ACTIVE_INST_VALU %.0f per launch = %.3f per VALU instruction;  GRBM_GUI_ACTIVE %.0f (summed over the 8 XCDs)" %
          (m["SQ_ACTIVE_INST_VALU"], m["SQ_ACTIVE_INST_VALU"] / max(m.get("SQ_INSTS_VALU", 1), 1), m["GRBM_GUI_ACTIVE"]))
