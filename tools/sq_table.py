"""One line per kernel from a tools/pmc_table.py CSV: issue rates, waves per CU, waits. usage: sq_table.py <csv>"""
import collections
import csv
import sys

d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["kernel"][:70]][r["counter"]] = float(r["mean_per_dispatch"])
for k, c in d.items():
    if "SQ_BUSY_CU_CYCLES" not in c:
        continue
    b = c["SQ_BUSY_CU_CYCLES"]
    w = max(c.get("SQ_WAVES", 1), 1)
    print(f"{k:72s} VALU {c['SQ_INSTS_VALU'] * 2 / (4 * b) * 100:5.1f}%  LDS {c['SQ_LDS_IDX_ACTIVE'] / b * 100:5.1f}% (conf "
          f"{c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1) * 100:4.1f}%)  waves/CU {c['SQ_WAVE_CYCLES'] * 4 / b:5.1f}  VALU/wave "
          f"{c['SQ_INSTS_VALU'] / w:7.0f}  LDS/wave {c['SQ_INSTS_LDS'] / w:6.0f}  SALU/wave {c.get('SQ_INSTS_SALU', 0) / w:6.0f}  wait_any "
          f"{c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES'] * 100:5.1f}%  wait_inst_any {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES'] * 100:5.1f}%  "
          f"wait_lds {c.get('SQ_WAIT_INST_LDS', 0) / c['SQ_WAVE_CYCLES'] * 100:5.1f}%")
