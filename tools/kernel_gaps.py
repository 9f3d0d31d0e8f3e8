import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
# last 12 kernels = the last repetitions
for a, b in zip(rows[-13:-1], rows[-12:]):
    print("%-42s dur %8.1f us   gap to next %7.1f us" % (a[2], (a[1] - a[0]) / 1e3, (b[0] - a[1]) / 1e3))
