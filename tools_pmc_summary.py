import csv, glob, collections, sys
suf = sys.argv[1]
for d in sorted(glob.glob('gpurun_out/pmc_%s_*/' % suf)):
    try:
        rows = list(csv.DictReader(open(d + 'pmc_counter_collection.csv')))
    except Exception:
        print(d, 'no data'); continue
    agg = collections.defaultdict(list)
    for r in rows:
        if 'wpt_pathtrace' in r['Kernel_Name'] and 'true>' not in r['Kernel_Name'].split('(')[0].replace('true, false','X') :
            agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'], r['VGPR_Count'], r['LDS_Block_Size'])].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print('%-42s %-26s vgpr=%s lds=%s n=%d mean=%.6g' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v)))
