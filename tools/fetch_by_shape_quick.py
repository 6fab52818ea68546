import sqlite3, glob, sys, csv, collections
d, cf = sys.argv[1], sys.argv[2]
db = glob.glob(d + "/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = [(n, float(v)) for _, n, v in cur.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name='FETCH_SIZE' order by dispatch_id") if "conv_igemm" in n]
L = list(csv.DictReader(open(cf)))
rows = rows[-len(L):]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r, (n, v) in zip(L, rows):
    k = (r["M"], r["N"], "grouped" if float(r["gflop"]) > 2.1 * int(r["M"]) * int(r["N"]) * int(r["K"]) / 1e9 else "single")
    a = agg[k]; a[0] += 1; a[1] += 2 * v * 1024 / 1e6; a[2] += float(r["alg_mbytes"])
tot_f = sum(a[1] for a in agg.values()); tot_a = sum(a[2] for a in agg.values())
print("total fetch MB/launch", round(tot_f / len(L), 1), "alg", round(tot_a / len(L), 1))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(k, a[0], "fetch/launch %.1f MB  alg %.1f MB" % (a[1] / a[0], a[2] / a[0]))
