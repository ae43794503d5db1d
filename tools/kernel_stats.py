import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 6]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(8))
