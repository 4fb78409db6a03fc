#!/usr/bin/env python3
"""Summarise ONE rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY + GRBM_GUI_ACTIVE; `--kernel-trace` only, as gpurun demands) into MFMA-pipe utilisation per kernel:

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)     busy cycles of the matrix pipe per SIMD over the cycles
                                                                                      the dispatch kept the chip active (MI355X_MICROARCH.md:
                                                                                      BUSY_CYCLES counts cycles summed over SIMDs)
  mfma_busy_of_wave_time = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)            (quad-cycles; the figure of profiles/r4_pmc_attn_summary.log)

Usage: pmc_mfma_busy.py <counter_collection.csv> [out.json]"""
import collections
import csv
import json
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name).strip()
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[name].add(r.get("Dispatch_Id") or r.get("Correlation_Id") or len(launches[name]))
rows = []
for k, c in acc.items():
    busy, gui, wave = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
    if busy <= 0 or gui <= 0:
        continue
    n = max(1, len(launches[k]))
    rows.append({"kernel": k, "launches": n, "mfma_busy": busy / 1024.0 / (gui / 8.0), "mfma_busy_of_wave_time": busy / (4.0 * wave) if wave else None,
                 "active_cycles_per_launch": gui / 8.0 / n, "mfma_insts_per_launch": c.get("SQ_INSTS_MFMA", 0.0) / n,
                 "wait_any_share": c.get("SQ_WAIT_ANY", 0.0) / wave if wave else None, "wait_inst_any_share": c.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None,
                 "active_inst_share": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else None})
rows.sort(key=lambda r: -r["active_cycles_per_launch"] * r["launches"])
for r in rows[:16]:
    print(f'{r["mfma_busy"]:.3f} busy  ({r["mfma_busy_of_wave_time"] or 0:.3f} of wave time)  n={r["launches"]:5d}  {r["active_cycles_per_launch"]/1e6:8.3f} M active cycles/launch  {r["kernel"][:80]}')
if len(sys.argv) > 2:
    json.dump({"note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8), one rocprofv3 --pmc pass (--kernel-trace only); see profiles/README.md",
               "kernels": rows[:40]}, open(sys.argv[2], "w"), indent=1)
