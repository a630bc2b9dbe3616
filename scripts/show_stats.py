import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
rows = list(csv.reader(open(f[0])))
for r in rows[1:int(sys.argv[2]) if len(sys.argv) > 2 else 32]:
    print(r[0][:72].ljust(72), r[1].rjust(6), r[3][:9].rjust(10), r[4][:6])
