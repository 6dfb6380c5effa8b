import csv,glob,sys
for d in sys.argv[1:]:
    f=glob.glob(d+'/*/*_kernel_stats.csv')[0]
    print("##",d)
    for r in csv.DictReader(open(f)):
        if 'dec_' in r['Name'] or 'scan_' in r['Name'] or 'idct' in r['Name']:
            print("  ",r['Name'][:64].ljust(64), r['Calls'].rjust(5), "%8.1f us" % (float(r['AverageNs'])/1000))
