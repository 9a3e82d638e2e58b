"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel dispatch."""
import csv, collections, sys
d = sys.argv[1]; pre = sys.argv[2]
rows = list(csv.DictReader(open('%s/%s_counter_collection.csv' % (d, pre))))
disp = collections.OrderedDict()
for r in rows:
    k = (int(r['Dispatch_Id']), r['Kernel_Name'], r['Grid_Size'])
    disp.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
tr = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open('%s/%s_kernel_trace.csv' % (d, pre)))}
seen = set()
for (di, name, grid), c in disp.items():
    if 'conv' not in name: continue
    short = name[name.find('conv'):name.find('(')][:60]
    if (short, grid) in seen: continue
    seen.add((short, grid))
    dur = tr.get(di, 0); g = c.get('GRBM_GUI_ACTIVE', 0); wc = c.get('SQ_WAVE_CYCLES', 1) or 1
    out = "%-62s grid %-8s %.3f ms clk %.2f GHz" % (short, grid, dur * 1e-6, g / 8 / dur if dur else 0)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c: out += " mfma_util %.2f" % (c['SQ_VALU_MFMA_BUSY_CYCLES'] * 8 / (g * 1024) if g else 0)
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_INST_CYCLES_VMEM_RD'):
        if k in c: out += " %s %.2f" % (k[3:].lower(), c[k] / wc)
    for k in ('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_ADDR_CONFLICT', 'SQ_LDS_UNALIGNED_STALL', 'SQ_INSTS_LDS', 'SQ_BUSY_CYCLES'):
        if k in c: out += " %s %.3g" % (k[3:].lower(), c[k])
    print(out)
# generic dump of any other counters
for (di, name, grid), c in disp.items():
    if 'conv' not in name: continue
    short = name[name.find('conv'):name.find('(')][:60]
    key = (short, grid, 'x')
    if key in seen: continue
    seen.add(key)
    dur = tr.get(di, 0)
    extra = {k: v for k, v in c.items() if k.startswith(('TCC', 'TCP', 'TA_', 'FETCH', 'WRITE'))}
    if extra:
        print("   ", grid, "%.3f ms" % (dur * 1e-6), {k: "%.4g" % v for k, v in extra.items()})
